"""Drop-in mirror of the reference's statistics-kernel functions.

Same names, argument meaning, return shapes and error behaviour as
``/root/reference/PolyFastA.py`` for the hot path, with
every per-site / per-codon scan executed by the HIP kernels of libpfa_hip.so:

    getvarsites(d, seqlen)            PolyFastA.py:252-261
    getsfs(var)                       PolyFastA.py:274-282
    getvarCDSsites(d, seqlen)         PolyFastA.py:284-315
    get_syn_nonsyn_cod_sites(cod)     PolyFastA.py:319-434
    nucleotide_diversity(var)         PolyFastA.py:485-492
    wattersons_theta(var)             PolyFastA.py:494-497
    jukes_cantor_correction(x)        PolyFastA.py:499-500
    polymorphism(var, seqlen, jc)     PolyFastA.py:502-520
    Dvar(var)                         PolyFastA.py:522-534
    syncodfreq(cod)                   PolyFastA.py:536-557
    readfasta(file, stdin)            PolyFastA.py:227-250
    all_same(items), no_var()         PolyFastA.py:580-581, 436-437

``var`` is the reference's list of site columns.  ``getvarsites`` returns a
:class:`VarSites` — a real ``list`` of columns that also remembers the integer
summary the GPU produced for it, so ``polymorphism(var, …)`` and ``getsfs(var)``
need no second scan.  Any other list of columns (e.g. the ``var_s`` / ``var_n``
subsets built at PolyFastA.py:169-170) is packed and scanned on the GPU when it
is passed in.  There is no CPU implementation of the scans in this package.
"""
from __future__ import annotations

import numpy as np

from . import _lib, fasta, stats
from .scan import scan_matrix

_E_OF = {"A": 0, "C": 1, "T": 2, "G": 3}   # device codon encoding: (ascii >> 1) & 3


class VarSites(list):
    """List of variable-site columns + the device's integer summary of them.  The summary carries a
    fingerprint of the columns it was computed from (`_fingerprint`): a caller that edits the list or
    a column in place gets a fresh scan, as the reference's pure functions would give it
    (PolyFastA.py:485-520 recompute from `var` on every call)."""
    summary = None   # dict(N, S, H, sfs, sfs_bad, n_mono, fp)


def _fingerprint(var):
    """(columns, CRC of every character) of a list of columns; None when an entry is not a string"""
    import zlib
    try:
        return len(var), zlib.crc32("\x01".join(map("\x00".join, var)).encode("utf-8", "surrogatepass"))
    except TypeError:
        return None


def _columns_to_matrix(var) -> np.ndarray:
    """uint8 [N, S] from a list of S columns of N one-character strings."""
    n = len(var[0])
    rows = []
    for col in var:
        if len(col) != n:
            raise ValueError("columns of unequal length")
        try:
            b = "".join(c if c else "\0" for c in col).encode("latin-1")
        except UnicodeEncodeError as exc:
            raise ValueError("characters above U+00FF cannot be packed one byte per base") from exc
        if len(b) != n:
            raise ValueError("every column entry must be a single character")
        rows.append(np.frombuffer(b, dtype=np.uint8))
    return np.ascontiguousarray(np.stack(rows, axis=0).T)


def _summary_of(var):
    """Integer summary of a list of columns, scanning on the GPU if needed."""
    s = getattr(var, "summary", None)
    if s is not None and s.get("fp") is not None and s["fp"] == _fingerprint(var):
        return s
    m = _columns_to_matrix(var)
    out = scan_matrix(m)
    r = out.results[0]
    return {"N": m.shape[0], "S_listed": len(var), "S": int(r["seg_sites"]), "H": int(r["H"]),
            "sfs": out.sfs(0).astype(np.int64), "sfs_bad": int(r["sfs_bad"]),
            "n_mono": len(var) - int(r["seg_sites"]), "fp": _fingerprint(var)}


def all_same(items):
    return all(x == items[0] for x in items)


def no_var():
    return "0,0,0,NA"


def readfasta(file, stdin):
    """dict header -> upper-case sequence, or 1 (after printing the reference's
    message) when the input is not FASTA."""
    from ._lib import PFA_ERR_UNSUPPORTED, PfaError
    import io
    import sys
    raw = sys.stdin.buffer.read() if stdin else None
    try:
        if stdin and b"\r" in raw.replace(b"\r\n", b""):
            # standard input is split at \n only (a bare carriage return is a character of its line there, not a line end as in a
            # file: cli.process_pipe)
            raise PfaError(PFA_ERR_UNSUPPORTED, "standard input holds a bare carriage return")
        aln = fasta.parse_fasta_bytes(raw) if stdin else fasta.read_fasta_file(file)
        d = None if aln is None else aln.to_dict()
    except PfaError as e:       # non-ASCII characters in sequence lines: the reference's own text reading (a character is a site)
        if e.code != PFA_ERR_UNSUPPORTED:
            raise
        if stdin:
            d = fasta.read_text_alignment(io.TextIOWrapper(io.BytesIO(raw), encoding=getattr(sys.stdin, "encoding", None) or "utf-8",
                                                           errors=getattr(sys.stdin, "errors", None) or "strict", newline="\n"))
        else:
            with open(file, "r") as fh:
                d = fasta.read_text_alignment(fh)
    if d is None:
        print(f"# file {file} is not FASTA!")
        return 1
    return d


def getvarsites(d, seqlen):
    """(pos, var): positions and columns holding more than one distinct character."""
    names = list(d.keys())
    m = fasta.matrix_from_dict(d, seqlen)
    if len(names) == 0 or seqlen == 0:
        return [], VarSites()
    out = scan_matrix(m, want_pos=True)
    pos = [int(p) for p in out.var_positions(0)]
    var = VarSites([d[x][p:p + 1] for x in names] for p in pos)
    r = out.results[0]
    var.summary = {"N": len(names), "S_listed": len(pos), "S": int(r["seg_sites"]), "H": int(r["H"]),
                   "sfs": out.sfs(0).astype(np.int64), "sfs_bad": int(r["sfs_bad"]), "n_mono": 0}
    var.summary["fp"] = _fingerprint(var)
    return pos, var


def getsfs(var):
    """Folded SFS over A/C/G/T alleles; IndexError, like the reference, when a
    column carries fewer than two of them."""
    n = len(var[0])
    s = _summary_of(var)
    if s["sfs_bad"] or s["n_mono"]:
        raise IndexError("list index out of range")
    sfs = [0 for _ in range(int(n / 2))]
    for k, v in enumerate(s["sfs"]):
        sfs[k] = int(v)
    return sfs


def nucleotide_diversity(var):
    s = _summary_of(var)
    return stats.pi_total(s["H"], len(var[0]))


def wattersons_theta(var):
    return stats.wattersons_theta_total(len(var), len(var[0]))


def Dvar(var):
    return stats.tajima_sd(len(var), len(var[0]))


def jukes_cantor_correction(x):
    return stats.jukes_cantor_correction(x)


def polymorphism(var, seqlen, jc):
    if len(var) == 0:
        return 0, 0, 0, "NA"
    s = _summary_of(var)
    return stats.finish_polymorphism(len(var), s["H"], len(var[0]), seqlen, jc)


def _codon_index(codon: str):
    if len(codon) != 3 or any(ch not in _E_OF for ch in codon):
        return None
    return 16 * _E_OF[codon[0]] + 4 * _E_OF[codon[1]] + _E_OF[codon[2]]


def syncodfreq(cod):
    idx = _codon_index(cod)
    if idx is None:
        raise KeyError(cod)
    return _lib.load().pfa_codon_syn_thirds(idx) / 3


def get_syn_nonsyn_cod_sites(cod):
    """``cod`` = [list of unique codons, column start]; returns (S, N) absolute
    positions (host evaluation of pfa_codon.h's classifier; the kernels run its
    wave-uniform form, checked against the same vectors by classify_codon_sets_on_device)."""
    mask = 0
    for c in cod[0]:
        idx = _codon_index(c)
        if idx is not None:
            mask |= 1 << idx
    bits = _lib.load().pfa_classify_codon_mask(mask)
    syn = [p + cod[1] for p in range(3) if (bits >> p) & 1]
    non = [p + cod[1] for p in range(3) if (bits >> (3 + p)) & 1]
    return syn, non


def classify_codon_sets_on_device(codon_sets):
    """The classification of ``get_syn_nonsyn_cod_sites`` for many codon sets at once, by the
    wave-uniform classifier the scan kernels run (bits 0-2 synonymous, bits 3-5
    non-synonymous positions per set)."""
    import numpy as np
    masks = np.zeros(len(codon_sets), dtype=np.uint64)
    for i, cods in enumerate(codon_sets):
        m = 0
        for c in cods:
            idx = _codon_index(c)
            if idx is not None:
                m |= 1 << idx
        masks[i] = m
    out = np.zeros(len(masks), dtype=np.uint32)
    _lib.init()
    _lib.check(_lib.load().pfa_classify_codon_masks_dev(_lib.ptr(masks), len(masks), _lib.ptr(out)))
    return out


def getvarCDSsites(d, seqlen):
    """(synonymous-site count, S positions, N positions, nstops, missingcodpos).
    Position lists are ascending (the reference's within-codon order differs;
    only float summation order could depend on it).  The count is the reference's own
    column-by-column floating-point accumulation (PolyFastA.py:307), replayed on the host
    from the codon sets the card returns per column (stats.syn_sites_replay) — on a two-decimal rounding tie whose
    digit the reference itself does not fix (the summation order of a column of three or more codons follows its hash
    seed) the value it most often has (stats.syn_sites_most_likely, tests/golden G15)."""
    m = fasta.matrix_from_dict(d, seqlen)
    if m.shape[0] == 0 or seqlen == 0:
        return 0.0, [], [], 0, 0
    out = scan_matrix(m, cds=True, want_sn=True, exact_syn=True)
    r = out.results[0]
    return (float(out.sites_s[0]), [int(p) for p in out.syn_positions(0)],
            [int(p) for p in out.non_positions(0)], int(r["nstops"]), 3 * int(r["missing_cols"]))
