"""Deterministic synthetic alignments.

This is the *specification* of the generator, written with numpy uint64
arithmetic; ``csrc/pfa_synth.hip`` implements the identical integer recipe on
the device (``pfa_synth_fill_dev``) and on the host (``pfa_synth_fill_host``)
so that bench-sized batches (BASELINE config 5: 50 GB of bases) never have to
exist as files.  Everything is integer arithmetic on splitmix64 hashes of
``(seed, alignment index, site)``; the only floating-point step is the
threshold table built once per ``(N, theta)`` in :func:`make_params`, which is
computed here and handed to the native code as integers.

Model: ancestral base uniform over ACGT (coding mode: ancestral codon uniform
over the 61 sense codons); a site segregates with probability theta * a1(N);
the derived-allele count k is drawn with P(k) ~ 1/k on 1..N-1 (neutral SFS);
carriers are the rows r with ((a*r + b) mod N) < k for a per-site random unit
a and shift b (exactly k carriers); about 1 % of segregating sites carry a
second derived allele on the first k//2 carriers.
"""
from __future__ import annotations

import itertools
import math
from dataclasses import dataclass

import numpy as np

_U64 = np.uint64
_GOLDEN = _U64(0x9E3779B97F4A7C15)
_MIX1 = _U64(0xBF58476D1CE4E5B9)
_MIX2 = _U64(0x94D049BB133111EB)
_ALN_MULT = _U64(0xD6E8FEB86659FD93)
_CODON_SALT = _U64(0xC0D0C0D0C0D0C0D1)
_INJECT_SALT = _U64(0x1A7EC7ED5EED5EED)

ASCII_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)

# sense codons in the 2-bit code A=0,C=1,G=2,T=3 (index = 16*b0 + 4*b1 + b2),
# derived from the standard code: stops are TAA, TAG, TGA.
_STOPS = {(3, 0, 0), (3, 0, 2), (3, 2, 0)}
SENSE_CODE = np.array(
    [(a, b, c) for a in range(4) for b in range(4) for c in range(4) if (a, b, c) not in _STOPS],
    dtype=np.uint8,
)
assert SENSE_CODE.shape == (61, 3)


def splitmix64(x):
    """splitmix64 finaliser on uint64 scalars/arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = np.asarray(x, dtype=_U64) + _GOLDEN
        z = (x ^ (x >> _U64(30))) * _MIX1
        z = (z ^ (z >> _U64(27))) * _MIX2
        return z ^ (z >> _U64(31))


@dataclass(frozen=True)
class SynthParams:
    """Integer tables shared by the numpy, host-C++ and device generators."""
    n_rows: int
    seg_thresh: int          # of 2**24
    tri_thresh: int          # of 2**10
    cdf: np.ndarray          # uint32[n_rows-1], thresholds of 2**24 for k = 1..N-1
    units: np.ndarray        # uint32[], integers in [1, N) coprime to N (or [1] for N == 1)


def make_params(n_rows: int, theta: float, tri_frac: float = 0.01) -> SynthParams:
    n = int(n_rows)
    if n < 1:
        raise ValueError("n_rows must be >= 1")
    a1 = sum(1.0 / i for i in range(1, n))
    p_seg = min(max(theta * a1, 0.0), 1.0)
    seg_thresh = int(p_seg * (1 << 24))
    acc, cdf = 0.0, []
    for k in range(1, n):
        acc += 1.0 / k
        cdf.append(min(int(acc / a1 * (1 << 24)), 1 << 24))
    if cdf:
        cdf[-1] = 1 << 24
    units = [u for u in range(1, max(n, 2)) if math.gcd(u, n) == 1] or [1]
    return SynthParams(
        n_rows=n,
        seg_thresh=seg_thresh,
        tri_thresh=int(tri_frac * 1024),
        cdf=np.asarray(cdf, dtype=np.uint32),
        units=np.asarray(units, dtype=np.uint32),
    )


def alignment_key(seed: int, aln_index: int) -> np.uint64:
    with np.errstate(over="ignore"):
        return splitmix64(_U64(seed) ^ (_U64(aln_index) * _ALN_MULT))


def synth_alignment(seed: int, aln_index: int, n_sites: int, params: SynthParams,
                    coding: bool = False) -> np.ndarray:
    """uint8 matrix [N, L] of ASCII bases for alignment `aln_index` of stream `seed`."""
    n, L = params.n_rows, int(n_sites)
    akey = alignment_key(seed, aln_index)
    with np.errstate(over="ignore"):
        site = np.arange(L, dtype=_U64)
        h1 = splitmix64(akey + site)
        h2 = splitmix64(h1)
        if coding:
            hc = splitmix64((akey ^ _CODON_SALT) + site // _U64(3))
            anc = SENSE_CODE[(hc % _U64(61)).astype(np.int64), (site % _U64(3)).astype(np.int64)]
        else:
            anc = (h1 & _U64(3)).astype(np.uint8)
        seg = ((h1 >> _U64(2)) & _U64(0xFFFFFF)) < _U64(params.seg_thresh)
        uk = ((h1 >> _U64(26)) & _U64(0xFFFFFF)).astype(np.uint32)
        d1 = (((h1 >> _U64(50)) & _U64(0xFF)) % _U64(3)).astype(np.uint8)
        unit = params.units[((h2 & _U64(0xFFFF)) % _U64(len(params.units))).astype(np.int64)].astype(_U64)
        shift = ((h2 >> _U64(16)) & _U64(0xFFFFFF)) % _U64(n)
        tri = ((h2 >> _U64(40)) & _U64(0x3FF)) < _U64(params.tri_thresh)
    if n < 2:
        seg = np.zeros(L, dtype=bool)
        k = np.zeros(L, dtype=np.uint64)
    else:
        k = (1 + np.searchsorted(params.cdf, uk, side="right")).astype(_U64)
        k = np.minimum(k, _U64(n - 1))
    rows = np.arange(n, dtype=_U64)[:, None]
    rank = (unit[None, :] * rows + shift[None, :]) % _U64(n)
    carrier = seg[None, :] & (rank < k[None, :])
    second = carrier & (tri & (k >= 2))[None, :] & (rank < (k // _U64(2))[None, :])
    der1 = (anc + 1 + d1) & 3
    der2 = (anc + 1 + (d1 + 1) % 3) & 3
    code = np.where(second, der2[None, :], np.where(carrier, der1[None, :], anc[None, :]))
    return ASCII_ACGT[code.astype(np.int64)]


_MANY_SALT = _U64(0x3A11E1E5C0FFEE01)
_SUBSET_SALT = _U64(0x5B5E7F00DFACADE5)


def many_allele_matrix(seed: int, n_rows: int, n_sites: int, alphabet: bytes) -> np.ndarray:
    """uint8 matrix [N, L] for the many-row fixtures (tests/golden G10): every site draws 1-5 alleles from `alphabet`
    (repeats merge) with cubed-random weights, so sites carry three and more alleles at frequencies from a handful of rows
    to nearly all of them — second-largest counts far above any fixed number of SFS bins, per-site N^2 - sum c^2 above 32
    bits once N passes ~92 000.  Integer arithmetic on splitmix64 hashes only (reproducible anywhere)."""
    n, L = int(n_rows), int(n_sites)
    syms = np.frombuffer(alphabet, dtype=np.uint8)
    out = np.empty((n, L), dtype=np.uint8)
    with np.errstate(over="ignore"):
        key = splitmix64(_U64(seed) ^ _MANY_SALT)
        rows = np.arange(n, dtype=_U64)
        for s in range(L):
            hs = splitmix64(key + _U64(s) * _ALN_MULT)
            k = 1 + int(hs % _U64(5))
            hj = splitmix64(hs + np.arange(1, 1 + k, dtype=_U64))
            al = syms[(hj % _U64(len(syms))).astype(np.int64)]
            x = (hj >> _U64(43)) & _U64(0x1FFFFF)                 # 21 bits: x^3 < 2^63
            cum = np.cumsum(x * x * x + _U64(1 << 20))
            v = splitmix64(splitmix64(hs ^ _MANY_SALT) + rows) % cum[-1]
            out[:, s] = al[np.searchsorted(cum, v, side="right")]
    return out


def row_subset(seed: int, n_rows: int, frac: float) -> np.ndarray:
    """ascending uint32 indices of about `frac` of the rows (hash-driven, reproducible): a population of a huge alignment"""
    with np.errstate(over="ignore"):
        h = splitmix64((_U64(seed) ^ _SUBSET_SALT) + np.arange(int(n_rows), dtype=_U64))
    return np.flatnonzero((h & _U64(0xFFFF)) < _U64(int(frac * 65536))).astype(np.uint32)


def inject_symbols(matrix: np.ndarray, seed: int, frac: float, alphabet: bytes) -> np.ndarray:
    """Overwrite about `frac` of the cells with symbols from `alphabet`
    (hash-driven, reproducible) — gaps, N, IUPAC codes, lower case."""
    out = matrix.copy()
    with np.errstate(over="ignore"):
        idx = np.arange(out.size, dtype=_U64)
        h = splitmix64((_U64(seed) ^ _INJECT_SALT) + idx)
    hit = (h & _U64(0xFFFF)) < _U64(int(frac * 65536))
    sym = np.frombuffer(alphabet, dtype=np.uint8)[((h >> _U64(16)) % _U64(len(alphabet))).astype(np.int64)]
    flat = out.reshape(-1)
    flat[hit] = sym[hit]
    return out


def plant_codons(matrix: np.ndarray, col: int, codons, rows=None) -> np.ndarray:
    """Write `codons` (3-letter strings) round robin into codon column `col` (sites 3 col .. 3 col + 2) of the rows `rows`
    (all rows by default): a column with 8 or 16 distinct codons puts the factor 8 into the denominator of the
    synonymous-site sum that makes two-decimal rounding ties possible (tests/golden G12)."""
    out = matrix.copy()
    rows = np.arange(out.shape[0]) if rows is None else np.asarray(rows)
    cod = np.array([list(c.encode()) for c in codons], dtype=np.uint8)
    out[rows, 3 * col:3 * col + 3] = cod[np.arange(len(rows)) % len(cod)]
    return out


def headers_for(n_rows: int, pops: int = 0):
    """``indiv{i}`` like the reference's examples, or ``pop{1+i%pops}_ind{i}``."""
    if pops:
        return [f"pop{1 + i % pops}_ind{i + 1}" for i in range(n_rows)]
    return [f"indiv{i + 1}" for i in range(n_rows)]


def to_dict(headers, matrix: np.ndarray):
    """dict header -> sequence string, the reference's in-memory alignment."""
    return {h: bytes(row).decode("ascii") for h, row in zip(headers, matrix)}


def to_fasta_text(headers, matrix: np.ndarray, wrap: int = 0, newline: str = "\n") -> str:
    parts = []
    for h, row in zip(headers, matrix):
        seq = bytes(row).decode("ascii")
        parts.append(">" + h + newline)
        if wrap:
            for i in range(0, len(seq), wrap):
                parts.append(seq[i:i + wrap] + newline)
        else:
            parts.append(seq + newline)
    return "".join(parts)


MIXED_MODES = ([], ["--cds"], ["--cds", "--jc"], ["--jc"])      # (the reference's own -p dies on every input under Python 3: G6 / G13 pin it per key)


def mixed_directory(seed: int):
    """A directory of small, awkward FASTA files and the flags to run on it: (dict file name -> bytes, list of flags).  For the
    randomised command-line fixtures (tests/golden G14: the reference's transcript is frozen per seed, the files are made again
    from the seed).  Per file: 1 to 30 sequences of 1 to 1 001 sites around a random consensus; then one of: as is, lower case
    in some rows, other symbols (N - n R Y * ?), unequal lengths, non-ASCII characters at one site, no header at all, blank
    lines and trailing blanks, a repeated header, a header with a description, a byte that is not UTF-8 (seldom).  Line ends
    LF / CRLF / CR, sequences whole or wrapped at 60 / 7 characters."""
    rng = np.random.default_rng([int(seed), 14])
    flags = list(MIXED_MODES[seed % len(MIXED_MODES)])
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    files = {}
    for i in range(int(rng.integers(3, 25))):
        n = int(rng.choice([1, 2, 2, 3, 4, 7, 12, 30]))
        L = int(rng.choice([1, 2, 3, 5, 6, 63, 64, 65, 127, 128, 129, 255, 256, 257, 600, 1001]))
        base = rng.choice(acgt, size=L)
        seqs = []
        for r in range(n):
            s = base.copy()
            k = int(rng.integers(0, L // 20 + 2))
            if k:
                s[rng.integers(0, L, size=k)] = rng.choice(acgt, size=k)
            seqs.append(s.tobytes().decode())
        heads = [f"s{r}_pop{1 + r % 2}" for r in range(n)]
        kind = int(rng.integers(0, 12))
        if kind == 1:
            seqs = [x.lower() if rng.random() < 0.5 else x for x in seqs]
        elif kind == 2:
            for r in range(n):
                if rng.random() < 0.5:
                    j = int(rng.integers(0, L))
                    seqs[r] = seqs[r][:j] + str(rng.choice(list("N-nRY*?"))) + seqs[r][j + 1:]
        elif kind == 3 and L > 3 and n > 1:
            seqs[-1] = seqs[-1][:-1]
        elif kind == 4:
            j = int(rng.integers(0, L))
            for r in range(n):
                if rng.random() < 0.5:
                    seqs[r] = seqs[r][:j] + str(rng.choice(list("éÉñЖ€"))) + seqs[r][j + 1:]
        elif kind == 6 and n > 1:
            heads[-1] = heads[0]
        elif kind == 7:
            heads = [h + " sample " + str(r) + ("  " if r % 2 else "") for r, h in enumerate(heads)]
        eol = str(rng.choice(["\n", "\n", "\n", "\r\n", "\r"]))
        wrap = int(rng.choice([0, 0, 60, 7]))
        if kind == 5:
            text = "no header here" + eol + seqs[0] + eol
        else:
            text = ""
            for r, x in enumerate(seqs):
                body = eol.join(x[k:k + wrap] for k in range(0, len(x), wrap)) if wrap else x
                if kind == 8:
                    body = body + "  " + eol
                text += f">{heads[r]}{eol}{body}{eol}"
            if kind == 8:
                text = eol + text + eol + eol
        blob = text.encode("utf-8")
        if kind == 9 and rng.random() < 0.15:
            blob = blob[:len(blob) // 2] + b"\xff" + blob[len(blob) // 2:]
        files[f"f{i:03d}.fa"] = blob
    return files, flags


WILD_FLAGS = ([], ["--cds"], ["--cds", "--jc"], ["--jc"])
_PALETTES = ("ACGT", "ACGTacgt", "ACGTN-", "ACGTNRYKMSWBDHV-?*.", "ACGT \t", "ACGT\x0b\x0c\x1c\x1d\x1e\x1f", "ACGT>", "ACGT0123456789,;:|", "ACGTU", "ACGTéñßЖıǆ", "AC")
_EOLS = ("\n", "\n", "\n", "\r\n", "\r", "\n\n", "\r\n\n", " \n", "\t\r\n", "\x0c\n", "\x1f\n")


def wild_directory(seed: int):
    """Directories further out than synth.mixed_directory: sites drawn from odd palettes (blanks and control white space INSIDE and
    at the end of lines, `>` in sequences, digits and punctuation, U, characters whose upper case is longer than they are), line ends
    mixed inside one file, lines before the first header, headers with leading blanks / commas / nothing at all, empty files,
    files of blank lines, a header as the last line without a line end, one or two sequences, identical sequences, codons of N and
    stops — and file names with blanks and commas.  (dict name -> bytes, flags)"""
    rng = np.random.default_rng([int(seed), 77])
    flags = list(WILD_FLAGS[seed % len(WILD_FLAGS)])
    files = {}
    for i in range(int(rng.integers(3, 16))):
        n = int(rng.choice([1, 2, 2, 3, 4, 5, 9, 20]))
        L = int(rng.choice([0, 1, 2, 3, 4, 6, 9, 12, 30, 31, 63, 66, 128, 129, 300]))
        pal = _PALETTES[int(rng.integers(0, len(_PALETTES)))]
        odd = float(rng.choice([0.0, 0.0, 0.02, 0.2]))
        base = [str(rng.choice(list("ACGT"))) for _ in range(L)]
        seqs = []
        for r in range(n):
            x = list(base)
            for j in rng.integers(0, max(L, 1), size=int(rng.integers(0, L // 10 + 2))).tolist():
                if L:
                    x[j] = str(rng.choice(list("ACGT")))
            if odd and L:
                for j in np.flatnonzero(rng.random(L) < odd).tolist():
                    x[j] = str(rng.choice(list(pal)))
            seqs.append("".join(x))
        if rng.random() < 0.1 and n > 1:
            seqs = [seqs[0]] * n
        hk = int(rng.integers(0, 8))
        heads = [(f"s{r}_pop{1 + r % 2}", f" s{r}", f"s{r},x={r}", f"s{r}\tdesc", "", f"s{r % 2}", f"s{r} pop1 ", f"ſ{r}é")[hk] for r in range(n)]
        kind = int(rng.integers(0, 14))
        eols = [str(rng.choice(list(_EOLS[:5]) if kind != 1 else list(_EOLS))) for _ in range(4 * n + 8)]
        e = itertools.cycle(eols)
        wrap = int(rng.choice([0, 0, 0, 60, 7, 1]))
        text = ""
        if kind == 2:
            text += "ACGT" + next(e) + "  stray line" + next(e)
        for r, x in enumerate(seqs):
            text += ">" + heads[r] + next(e)
            if wrap and x:
                for k in range(0, len(x), wrap):
                    text += x[k:k + wrap] + next(e)
            else:
                text += x + next(e)
            if kind == 3 and r == 0:
                text += next(e) + "   " + next(e)
        if kind == 4:
            text = ""
        elif kind == 5:
            text = "\n\n\r\n \n"
        elif kind == 6:
            text += ">last header without a line end"
        elif kind == 7:
            text = text.rstrip("\r\n \t\x0c\x1f")
        elif kind == 8 and n > 1 and L > 2:
            text = text.replace(seqs[-1], seqs[-1][:-1], 1)
        elif kind == 9:
            text = " " + text
        name = (f"w{i:02d}.fa", f"w{i:02d} x.fa", f"w{i:02d},y.fasta", f"w{i:02d}.é.fa")[int(rng.integers(0, 4))] if rng.random() < 0.3 else f"w{i:02d}.fa"
        files[name] = text.encode("utf-8")
    return files, flags


def big_directory(seed: int):
    """Fewer, larger alignments: 65 to 300 sequences (the 128-site-tile and the row-chunked kernels; synth.mixed_directory stays
    below 31), 64 to 1 200 sites, up to 15 % of a row's sites changed (columns of many alleles, codon columns of many codons), N / - /
    IUPAC / lower case sprinkled in."""
    rng = np.random.default_rng([int(seed), 99])
    flags = list(WILD_FLAGS[seed % len(WILD_FLAGS)])
    files = {}
    for i in range(int(rng.integers(2, 6))):
        n = int(rng.choice([65, 100, 130, 213, 214, 300]))
        L = int(rng.choice([64, 129, 255, 257, 600, 999, 1200]))
        base = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
        m = np.repeat(base[None, :], n, axis=0)
        rate = float(rng.choice([0.0005, 0.005, 0.05, 0.15]))
        hit = rng.random((n, L)) < rate
        m[hit] = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(hit.sum()))
        if rng.random() < 0.5:
            odd = rng.random((n, L)) < 0.002
            m[odd] = rng.choice(np.frombuffer(b"N-RYacgtn?", dtype=np.uint8), size=int(odd.sum()))
        eol = str(rng.choice(["\n", "\n", "\r\n"]))
        wrap = int(rng.choice([0, 60]))
        text = ""
        for r in range(n):
            x = m[r].tobytes().decode()
            body = eol.join(x[k:k + wrap] for k in range(0, L, wrap)) if wrap else x
            text += f">s{r}_pop{1 + r % 2}{eol}{body}{eol}"
        files[f"b{i:02d}.fa"] = text.encode()
    return files, flags
