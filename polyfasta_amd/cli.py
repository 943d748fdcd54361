"""Command line of polyfasta_amd: the PolyFastA.py surface (``--file/--dir/--pops/
--out/--pipe/--cds/--silent/--name/--jc``, same CSV columns and comment lines),
driven by batched GPU scans.

Mirrors ``main`` (PolyFastA.py:13-142), ``print_result`` / ``no_header``
(:147-198) and ``print_header`` (:211-225).  Differences, all deliberate:
  * files are parsed, packed and scanned in batches (one kernel launch per
    batch) instead of one at a time; output order is unchanged (sorted paths x
    population order as typed);
  * ``--pops`` works: the reference crashes on Python 3 at :126
    (``len(filter(...))``); the intended behaviour is implemented (App. A-1 of
    SURVEY.md): substring match of each key in the headers, one row per key;
  * every scan runs on the GPU; there is no CPU path;
  * two extra flags that do not change the default output: ``--gpus N`` (one process per
    GPU, files split by size, records gathered with RCCL) and ``--sfs`` (appends the folded
    site-frequency spectrum of ``getsfs``, PolyFastA.py:274-282, which the reference's epilog
    advertises but never prints).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from . import fasta, stats
from ._lib import PFA_ERR_IO, PFA_FILE_NOT_FASTA, PFA_FILE_UNEQUAL, PFA_FILE_UNSUPPORTED
from .scan import Batch, scan_batch

HEADER_PLAIN = "file,seqlen,pop,N,seg_sites,pi,theta,tajimasD"
HEADER_CDS = ("file,sites_S,sites_N,pop,N,seg_sites_S,seg_sites_N,pi_S,pi_N,theta_S,theta_N,"
              "tajimasD_S,tajimasD_N,nstops")
NO_VAR = "0,0,0,NA"

# field order of pfa_result / RESULT_DTYPE
_H, _HS, _HN, _SEG, _BAD, _SEGS, _SEGN, _STOPS, _MISSING = range(9)


def _fields(res):
    """A device record as a tuple of Python ints (numpy records convert here; the batch
    path converts whole result arrays at once)."""
    return res if isinstance(res, tuple) else res.tolist()


def format_plain_row(name, pop, n_rows, seqlen, res, jc):
    """One non-coding CSV row from a device record (PolyFastA.py:182-198)."""
    r = _fields(res)
    s = r[_SEG]
    if s == 0:
        return f"{name},{seqlen},{pop},{n_rows},{NO_VAR}"
    p = stats.finish_polymorphism(s, r[_H], n_rows, seqlen, jc)
    return f"{name},{seqlen},{pop},{n_rows},{p[0]},{p[1]},{p[2]},{p[3]}"


def format_cds_row(name, pop, n_rows, seqlen, res, syn_num, jc, sites_s=None):
    """One coding CSV row (PolyFastA.py:151-180), including the reference's
    mis-ordered no-variation row (:160-162).  `sites_s`: the synonymous-site count as the reference
    accumulates it, for a unit on a rounding tie (scan.resolve_syn_ties); else it comes from the bins."""
    r = _fields(res)
    ssites = stats.syn_sites_from_bins(syn_num) if sites_s is None or sites_s != sites_s else float(sites_s)
    nsites = (seqlen - 3 * r[_MISSING]) - ssites
    if r[_SEG] == 0:
        return f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{n_rows},{NO_VAR},{NO_VAR},{0}"
    ps = stats.finish_polymorphism(r[_SEGS], r[_HS], n_rows, ssites, jc)
    pn = stats.finish_polymorphism(r[_SEGN], r[_HN], n_rows, nsites, jc)
    return (f"{name},{round(ssites, 2)},{round(nsites, 2)},{pop},{n_rows},{ps[0]},{pn[0]},{ps[1]},{pn[1]},"
            f"{ps[2]},{pn[2]},{ps[3]},{pn[3]},{r[_STOPS]}")


class Sink:
    """stdout, or append-to-file with the reference's progress chatter.  The reference
    re-opens the output file in append mode for every line (PolyFastA.py:156,175,185,193);
    here the lines are collected and appended in large writes, which leaves the same
    file behind at 100 k rows without 100 k opens."""

    FLUSH_LINES = 8192

    def __init__(self, out, silent):
        self.out, self.silent = out, silent
        self._pending = []

    def header(self, cds, sfs=False):
        if self.silent:
            return
        text = (HEADER_CDS if cds else HEADER_PLAIN) + (",sfs" if sfs else "")
        if self.out:
            with open(self.out, "w") as o:      # PolyFastA.py:103 -> :215/:222 ("w")
                o.write(text + "\n")
        else:
            print(text)

    def _write(self, text):
        self._pending.append(text)
        if len(self._pending) >= self.FLUSH_LINES:
            self.flush()

    def flush(self):
        if self.out and self._pending:
            with open(self.out, "a") as o:
                o.write("\n".join(self._pending) + "\n")
        self._pending = []

    def comment(self, text):
        if self.out:
            self._write(text)
        else:
            print(text)

    def bulk_ok(self):
        """a block of rows can go out in one piece when no per-row progress line is due"""
        return not self.out or self.silent

    def rows(self, block):
        """`block`: rows separated and terminated by newlines"""
        if self.out:
            self._write(block[:-1])
        else:
            sys.stdout.write(block)

    def rows_with_progress(self, block, names, pops):
        """--out without --silent, a block of rows at once: the same bytes on stdout (one progress
        piece per row, PolyFastA.py:159,177,186,195) and in the file as row-by-row writing leaves"""
        sys.stdout.write("".join(f"\r Writting to {self.out}, pop: {p:<10s}, parsing: {n:<15s}" for n, p in zip(names, pops)))
        sys.stdout.flush()
        self._write(block[:-1])

    def row(self, text, name, pop):
        if self.out:
            if not self.silent:
                print("\r", f"Writting to {self.out}, pop: {pop:<10s}, parsing: {name:<15s}", end="", flush=True)
            self._write(text)
        else:
            print(text)


def build_parser():
    p = argparse.ArgumentParser(
        prog="polyfasta_amd", formatter_class=argparse.RawTextHelpFormatter,
        description="GPU (MI355X) estimator of nucleotide diversity (theta_pi), Watterson's theta (theta_w)\n"
                    "and Tajima's D for coding and non-coding alignments; PolyFastA.py-compatible.",
        epilog="Examples:\n"
               "  python -m polyfasta_amd -f aln.fas -p pop1,pop2        rows per population (header substring)\n"
               "  python -m polyfasta_amd -f aln.fas -p pop1,pop2 --cds --jc\n"
               "  python -m polyfasta_amd -d fastaDir/ --out all.csv\n"
               "  python -m polyfasta_amd --pipe -p pop1,pop2 < aln.fas\n")
    p.add_argument("--file", "-f", nargs="*", type=str, default="", help="one or several alignment files in FASTA format.")
    p.add_argument("--dir", "-d", type=str, default=".", help="directory containing FASTA files only.")
    p.add_argument("--pops", "-p", nargs="?", default=False, metavar="pop1,pop2,pop3", type=str,
                   help="split alignment by populations: comma-separated strings found in the sequence headers.")
    p.add_argument("--out", "-o", type=str, default="", help="name of output file (default: print to screen).")
    p.add_argument("--pipe", "-i", action="store_true", default=False, help="read one FASTA alignment from STDIN.")
    p.add_argument("--cds", "-c", action="store_true", default=False,
                   help="the alignment is protein coding: split into synonymous and nonsynonymous sites.")
    p.add_argument("--silent", "-s", action="store_true", default=False, help="suppress header and verbose output.")
    p.add_argument("--name", "-n", type=str, default=False, help="name of the DNA region to show in output.")
    p.add_argument("--jc", action="store_true", default=False, help="Jukes-Cantor correction for Pi.")
    p.add_argument("--gpus", type=int, default=0, metavar="N",
                   help="(polyfasta_amd) scan on N GPUs of this node: one process per GPU, the files split by size.")
    p.add_argument("--sfs", action="store_true", default=False,
                   help="(polyfasta_amd) append a column `sfs`: the folded site-frequency spectrum, bins 1..N/2 joined by ';'.")
    return p


def _events_for(fidx, path, name, status, names, n_rows, seqlen, args, add_unit):
    """Output events of one parsed file (PolyFastA.py:107-140); `add_unit(rows)`
    registers a scan unit and returns its index."""
    evs = []
    if status == PFA_FILE_NOT_FASTA:
        return [(fidx, "stdout", f"# file {path} is not FASTA!")]                   # :247 (always stdout)
    if status == PFA_FILE_UNEQUAL:
        return [(fidx, "comment", f"# Sequences do not have the same length: {name}")]   # :135-140
    if status == PFA_FILE_UNSUPPORTED:     # this file only: the rest of the batch is scanned (deviations)
        return [(fidx, "comment", f"# Sequences hold non-ASCII characters (not supported, one byte per base): {name}")]
    if args.cds and seqlen % 3 != 0:                                               # :114-119
        evs.append((fidx, "comment", f"# CDS sequence length is not a multiple of 3: {name}"))
    if not args.pops:                                                              # :120-121
        evs.append((fidx, "unit", add_unit(None), name, "NA", n_rows, seqlen))
    else:
        for pop in args.pops.split(","):                                           # :123-134
            # header names arrive as bytes from the native loader (str from the --pipe parser):
            # substring match on the same kind
            key = pop.encode("utf-8", "surrogateescape") if names and isinstance(names[0], bytes) else pop
            rows = [i for i, h in enumerate(names) if key in h]
            if len(rows) == 0:
                evs.append((fidx, "comment", f"# Pop {pop} string was not found in fasta headers."))
                continue
            evs.append((fidx, "unit", add_unit(rows), name, pop, len(rows), seqlen))
    return evs


def _attach(pending, out, done, want_sfs=False):
    """Unit events get their device record, their bins of synonymous-site numerators and — with
    --sfs — their folded SFS (None when the reference's getsfs would raise: a variable column
    with fewer than two A/C/G/T alleles, PolyFastA.py:279-281)."""
    recs = out.results.tolist() if out is not None else None     # tuples of Python ints, one conversion per batch
    for ev in pending:
        if ev[1] == "unit":
            fidx, _, unit, name, pop, n_rows, seqlen = ev
            syn = out.syn_num[unit] if out.syn_num is not None else None
            sfs = None
            if want_sfs and recs[unit][_BAD] == 0:
                sfs = out.sfs(unit).tolist()
            done.append((fidx, "unit", name, pop, n_rows, seqlen, recs[unit], syn, sfs,
                         None if out.sites_s is None else float(out.sites_s[unit])))
        else:
            done.append(ev)


def text_alignment_units(data, path, args, keys):
    """A file whose sequence lines hold non-ASCII characters (the native loader keeps one byte per base and hands such files back),
    from the reference's own reading of it (`fasta.read_text_alignment`): every distinct non-ASCII character gets a byte value of
    its own — the scan only asks whether two rows hold the same symbol and whether a symbol is exactly A, C, G or T — and
    the matrix is scanned on the card as one small batch.  Returns (units, lines): units as (key index or PFA_NO_KEY,
    n_rows, n_sites, record, syn row, sfs list, sites_S override or None), lines as (sub-order, kind, text); or None when the file holds more
    distinct characters than a byte has values (fasta.byte_codes).  Sequences of no sites at all are an alignment like any other
    (a piped file with bare-CR line ends is one header line: one sequence, zero sites)."""
    from ._lib import PFA_NO_KEY
    name = path.split("/")[-1]
    if data is None:
        return [], [(0, "stdout", f"# file {path} is not FASTA!")]                                      # :247
    names, seqs = list(data.keys()), list(data.values())
    if len(set(len(x) for x in seqs)) != 1:
        return [], [(0, "comment", f"# Sequences do not have the same length: {name}")]                # :135-140
    seqlen = len(seqs[0])
    table = fasta.byte_codes(seqs)
    if table is None:
        return None
    m = np.frombuffer("".join(seqs).translate(table).encode("latin-1"), dtype=np.uint8).reshape(len(seqs), seqlen)
    lines = []
    if args.cds and seqlen % 3 != 0:                                                                    # :114-119
        lines.append((0, "comment", f"# CDS sequence length is not a multiple of 3: {name}"))
    batch = Batch(stride_align=128)
    mid = batch.add_matrix(m)
    heads = []
    if not keys:
        batch.add_unit(mid, cds=args.cds)
        heads.append((PFA_NO_KEY, len(seqs)))
    else:
        for j, key in enumerate(keys):                                                                  # :123-134
            rows = [i for i, h in enumerate(names) if key in h]
            if not rows:
                lines.append((1 + j, "comment", f"# Pop {key} string was not found in fasta headers."))
                continue
            batch.add_unit(mid, rows=np.asarray(rows, dtype=np.uint32), cds=args.cds)
            heads.append((j, len(rows)))
    units = []
    if heads:
        out = scan_batch(batch)
        for i, (j, n_rows) in enumerate(heads):
            bad = int(out.results[i]["sfs_bad"]) != 0
            units.append((j, n_rows, seqlen, out.results[i], None if out.syn_num is None else out.syn_num[i],
                          None if (bad or not args.sfs) else out.sfs(i).tolist(),
                          None if out.sites_s is None else float(out.sites_s[i])))
    return units, lines


def process_pipe(indexed_paths, args):
    """--pipe: one alignment from standard input (PolyFastA.py:96-100, 230-236), parsed by the
    native packer, scanned as one small batch; returns its output events."""
    done = []
    for fidx, path in indexed_paths:
        from ._lib import PFA_ERR_UNSUPPORTED, PfaError
        try:
            raw = sys.stdin.buffer.read()
            if not raw.isascii():      # the reference reads TEXT (sys.stdin.readlines(), :231): what a byte >= 0x80 is, or whether
                raise PfaError(PFA_ERR_UNSUPPORTED, "standard input holds bytes that are not ASCII")       # it kills the run, is the text layer's call
            if b"\r" in raw.replace(b"\r\n", b""):
                # a carriage return that is not part of \r\n: a line end in a FILE (open(file, 'r'): universal newlines, what the native
                # parser implements), but not on standard input, which Python splits at \n only — there it is a character of its line
                # (found by tools/g14_explore.py: a file with bare-CR line ends is one header line when piped)
                raise PfaError(PFA_ERR_UNSUPPORTED, "standard input holds a bare carriage return")
            aln = fasta.parse_fasta_bytes(raw, 128)
            status = PFA_FILE_NOT_FASTA if aln is None else (0 if aln.same_length else PFA_FILE_UNEQUAL)
        except PfaError as e:
            if e.code != PFA_ERR_UNSUPPORTED:
                raise
            aln, status = None, PFA_FILE_UNSUPPORTED
            import io
            text = io.TextIOWrapper(io.BytesIO(raw), encoding=getattr(sys.stdin, "encoding", None) or "utf-8",
                                    errors=getattr(sys.stdin, "errors", None) or "strict", newline="\n")   # sys.stdin.readlines(), :231: lines end at \n only
            keys = args.pops.split(",") if args.pops else []
            got = text_alignment_units(fasta.read_text_alignment(text), path, args, keys)
            if got is not None:      # non-ASCII characters in sequence lines: read as text, the way the reference reads it
                from ._lib import PFA_NO_KEY
                name = path.split("/")[-1]
                items = [(sub, 0, (fidx, kind, line)) for sub, kind, line in got[1]]
                items += [(0 if j == PFA_NO_KEY else 1 + j, 1,
                           (fidx, "unit", name, "NA" if j == PFA_NO_KEY else keys[j], n_rows, seqlen, tuple(rec.tolist()), syn, sfs, ssx))
                          for j, n_rows, seqlen, rec, syn, sfs, ssx in got[0]]
                items.sort(key=lambda t: t[:2])
                done.extend(t[2] for t in items)
                continue
        batch = Batch(stride_align=128)
        mid = batch.add_matrix(aln.matrix) if status == 0 else None
        pending = _events_for(fidx, path, path.split("/")[-1], status, aln.names if aln else None,
                              aln.n_rows if aln else 0, int(aln.seq_lens[0]) if status == 0 else 0, args,
                              lambda rows: batch.add_unit(mid, rows=None if rows is None else np.asarray(rows, dtype=np.uint32), cds=args.cds))
        _attach(pending, scan_batch(batch) if batch.n_units else None, done, args.sfs)
    return done


def format_plain_rows_native(units, jc):
    """The non-coding rows of many units in one call into the library
    (csrc/pfa_format.cpp: polymorphism() and repr(float) reproduced bit for bit).
    `units`: sequences (name, pop, n_rows, seqlen, record tuple).  Returns the rows as one
    text block ending in a newline, or None when some row needs the Python path (the
    reference would raise there)."""
    import ctypes
    from . import _lib
    n = len(units)
    prefixes = [f"{u[0]},{u[3]},{u[1]},{u[2]}," for u in units]
    blob = "".join(prefixes).encode()
    lens = np.fromiter(map(len, prefixes), dtype=np.uint64, count=n)
    if int(lens.sum()) != len(blob):                       # non-ASCII names: lengths in bytes
        lens = np.fromiter((len(x.encode()) for x in prefixes), dtype=np.uint64, count=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    seg = np.fromiter((u[4][_SEG] for u in units), dtype=np.uint32, count=n)
    hsum = np.fromiter((u[4][_H] for u in units), dtype=np.uint64, count=n)
    nrows = np.fromiter((u[2] for u in units), dtype=np.uint32, count=n)
    seqlen = np.fromiter((u[3] for u in units), dtype=np.float64, count=n)
    uniq, inv = np.unique(nrows, return_inverse=True)
    tab = np.array([((stats.harmonic_terms(int(k))[0],) + stats.tajima_e(int(k))) if k >= 2 else (0.0, 0.0, 0.0)
                    for k in uniq], dtype=np.float64).reshape(-1, 3)
    a1, e1, e2 = (np.ascontiguousarray(tab[inv, c]) for c in range(3))
    out = np.empty(len(blob) + 128 * n + 16, dtype=np.uint8)
    bad = ctypes.c_int64(-1)
    got = _lib.load().pfa_format_plain_rows(blob, _lib.ptr(off), n, _lib.ptr(seg), _lib.ptr(hsum), _lib.ptr(nrows),
                                            _lib.ptr(seqlen), _lib.ptr(a1), _lib.ptr(e1), _lib.ptr(e2), int(bool(jc)),
                                            _lib.ptr(out), len(out), ctypes.byref(bad))
    if got < 0:
        if bad.value >= 0:
            return None
        _lib.check(got)
    return out[:got].tobytes().decode()


def _blob(parts):
    """(utf-8 bytes of the concatenation, uint64 offsets) of a list of strings"""
    n = len(parts)
    blob = "".join(parts).encode()
    lens = np.fromiter(map(len, parts), dtype=np.uint64, count=n)
    if int(lens.sum()) != len(blob):                       # non-ASCII text: lengths in bytes
        lens = np.fromiter((len(x.encode()) for x in parts), dtype=np.uint64, count=n)
    off = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    return blob, off


def _sample_size_constants(nrows):
    uniq, inv = np.unique(nrows, return_inverse=True)
    tab = np.array([((stats.harmonic_terms(int(k))[0],) + stats.tajima_e(int(k))) if k >= 2 else (0.0, 0.0, 0.0)
                    for k in uniq], dtype=np.float64).reshape(-1, 3)
    return tuple(np.ascontiguousarray(tab[inv, c]) for c in range(3))


def format_cds_rows_native(units, jc):
    """The coding rows of many units in one call (csrc/pfa_format.cpp).  `units`: sequences
    (name, pop, n_rows, seqlen, record tuple, syn_num row, sites_S override or None).  Returns the text
    block, or None when some row needs the Python path."""
    import ctypes
    from . import _lib
    from ._lib import RESULT_DTYPE
    n = len(units)
    names, noff = _blob([f"{u[0]}," for u in units])
    mids, moff = _blob([f",{u[1]},{u[2]}," for u in units])
    rec = np.array([u[4] for u in units], dtype=RESULT_DTYPE)
    syn = np.ascontiguousarray(np.stack([np.asarray(u[5], dtype=np.uint32) for u in units]))
    ssx = np.array([np.nan if len(u) < 7 or u[6] is None else u[6] for u in units], dtype=np.float64)
    nrows = np.fromiter((u[2] for u in units), dtype=np.uint32, count=n)
    seqlen = np.fromiter((u[3] for u in units), dtype=np.uint32, count=n)
    a1, e1, e2 = _sample_size_constants(nrows)
    out = np.empty(len(names) + len(mids) + 320 * n + 16, dtype=np.uint8)
    bad = ctypes.c_int64(-1)
    got = _lib.load().pfa_format_cds_rows(names, _lib.ptr(noff), mids, _lib.ptr(moff), n, _lib.ptr(rec), _lib.ptr(syn),
                                          _lib.ptr(ssx), _lib.ptr(nrows), _lib.ptr(seqlen), _lib.ptr(a1), _lib.ptr(e1), _lib.ptr(e2),
                                          int(bool(jc)), _lib.ptr(out), len(out), ctypes.byref(bad))
    if got < 0:
        if bad.value >= 0:
            return None
        _lib.check(got)
    return out[:got].tobytes().decode()


NATIVE_ROWS_MIN = 64      # below this the per-row Python path is as fast


def _sfs_text(sfs):
    return "NA" if sfs is None else ";".join(map(str, sfs))


def emit(events, sink, cds, jc, sfs=False):
    """Finish the floats and write the rows / comment lines in order."""
    i, n = 0, len(events)
    while i < n:
        ev = events[i]
        if ev[1] == "comment":
            sink.comment(ev[2])
            i += 1
        elif ev[1] == "stdout":
            print(ev[2])
            i += 1
        else:
            j = i
            while j < n and events[j][1] == "unit":
                j += 1
            run = events[i:j]
            block = None
            if len(run) >= NATIVE_ROWS_MIN and all(isinstance(e[5], int) and 0 <= e[5] < 2 ** 32 for e in run):
                if not cds:
                    block = format_plain_rows_native([(e[2], e[3], e[4], e[5], _fields(e[6])) for e in run], jc)
                elif all(e[7] is not None for e in run):
                    block = format_cds_rows_native([(e[2], e[3], e[4], e[5], _fields(e[6]), e[7], e[9] if len(e) > 9 else None) for e in run], jc)
            if block is not None and sink.bulk_ok() and not sfs:
                sink.rows(block)
            else:
                rows = block.split("\n") if block is not None else None
                for k, e in enumerate(run):
                    name, pop, n_rows, seqlen, res, syn = e[2:8]
                    text = rows[k] if rows is not None else (
                        format_cds_row(name, pop, n_rows, seqlen, res, syn, jc, e[9] if len(e) > 9 else None) if cds
                        else format_plain_row(name, pop, n_rows, seqlen, res, jc))
                    if sfs:
                        text += "," + _sfs_text(e[8])
                    sink.row(text, name, pop)
            i = j
    sink.flush()


# ---------------------------------------------------------------- row blocks
# What a run of scanned files turns into, as arrays: one entry per CSV row-to-be (file index, key
# index, N, alignment length, device record, bins of synonymous-site numerators), plus the rare
# lines that are not rows ("specials": comment lines and the always-on-stdout "is not FASTA").
# Nothing per file happens in Python while a block is regular.
META_DTYPE = np.dtype([("fidx", "<u4"), ("key", "<u4"), ("n_rows", "<u4"), ("n_sites", "<u4")])


class RowBlock:
    def __init__(self, meta, rec, syn=None, sfs=None, specials=None, io_error=None, ssx=None):
        self.meta, self.rec, self.syn = meta, rec, syn
        # coding rows: NaN, or the synonymous-site count of a unit on a rounding tie as the reference accumulates it (scan.resolve_syn_ties)
        self.ssx = ssx if ssx is not None or syn is None else np.full(len(meta), np.nan, dtype=np.float64)
        self.sfs = sfs                    # per unit: list of bins, or None (--sfs only)
        self.specials = specials or []    # (fidx, sub, kind, text); sub orders the lines of one file
        self.io_error = io_error          # smallest index of a file that could not be read (the job dies THERE, PolyFastA.py:239), or None

    @staticmethod
    def concat(blocks, coding):
        from ._lib import PFA_SYN_BINS, RESULT_DTYPE
        blocks = [b for b in blocks if b is not None]
        meta = np.concatenate([b.meta for b in blocks]) if blocks else np.zeros(0, dtype=META_DTYPE)
        rec = np.concatenate([b.rec for b in blocks]) if blocks else np.zeros(0, dtype=RESULT_DTYPE)
        syn = None
        if coding:
            syn = (np.concatenate([b.syn if b.syn is not None else np.zeros((len(b.meta), PFA_SYN_BINS), dtype=np.uint32) for b in blocks])
                   if blocks else np.zeros((0, PFA_SYN_BINS), dtype=np.uint32))
        sfs = None
        if any(b.sfs is not None for b in blocks):
            sfs = [x for b in blocks for x in (b.sfs if b.sfs is not None else [None] * len(b.meta))]
        ssx = None
        if coding:
            ssx = (np.concatenate([b.ssx if b.ssx is not None else np.full(len(b.meta), np.nan) for b in blocks])
                   if blocks else np.zeros(0, dtype=np.float64))
        return RowBlock(meta, rec, syn, sfs, [x for b in blocks for x in b.specials], ssx=ssx)

    def sorted(self):
        """file order, then population order as typed (PolyFastA.py:104, :124)"""
        from ._lib import PFA_NO_KEY
        sub = np.where(self.meta["key"] == PFA_NO_KEY, 0, self.meta["key"]).astype(np.int64)
        order = np.lexsort((sub, self.meta["fidx"]))
        if (order == np.arange(len(order))).all():
            return self
        return RowBlock(self.meta[order], self.rec[order], None if self.syn is None else self.syn[order],
                        None if self.sfs is None else [self.sfs[i] for i in order.tolist()], self.specials,
                        ssx=None if self.ssx is None else self.ssx[order])

    def split_below(self, fidx_end):
        """(rows and special lines of files < fidx_end, the rest); self is sorted"""
        k = int(np.searchsorted(self.meta["fidx"], fidx_end, side="left"))
        lo = [x for x in self.specials if x[0] < fidx_end]
        hi = [x for x in self.specials if x[0] >= fidx_end]
        part = lambda a, b: RowBlock(self.meta[a:b], self.rec[a:b], None if self.syn is None else self.syn[a:b],   # noqa: E731
                                     None if self.sfs is None else self.sfs[a:b], None, ssx=None if self.ssx is None else self.ssx[a:b])
        first, rest = part(0, k), part(k, len(self.meta))
        first.specials, rest.specials = lo, hi
        return first, rest

    def to_bytes(self):
        import pickle
        return pickle.dumps((self.meta.tobytes(), self.rec.tobytes(), None if self.syn is None else self.syn.tobytes(),
                             self.sfs, self.specials, None if self.ssx is None else self.ssx.tobytes()), protocol=pickle.HIGHEST_PROTOCOL)

    @staticmethod
    def from_bytes(blob):
        import pickle
        from ._lib import PFA_SYN_BINS, RESULT_DTYPE
        meta, rec, syn, sfs, specials, ssx = pickle.loads(blob)
        return RowBlock(np.frombuffer(meta, dtype=META_DTYPE).copy(), np.frombuffer(rec, dtype=RESULT_DTYPE).copy(),
                        None if syn is None else np.frombuffer(syn, dtype=np.uint32).reshape(-1, PFA_SYN_BINS).copy(), sfs, specials,
                        ssx=None if ssx is None else np.frombuffer(ssx, dtype=np.float64).copy())


def block_from_batch(b, res, syn, sfs_flat, args, paths, keys):
    """The rows and special lines of one ingest batch (PolyFastA.py:107-140).  Irregular files —
    not FASTA, unequal lengths, non-ASCII sequence bytes, a coding length that is not a multiple of
    3, a population key without a match — are the only ones visited one by one."""
    from ._lib import PFA_NO_KEY
    nu = b.n_units
    meta = np.zeros(nu, dtype=META_DTYPE)
    meta["fidx"] = b.file_idx[b.unit_file]
    meta["key"] = b.unit_key
    meta["n_rows"] = b.unit_rows
    meta["n_sites"] = b.n_sites[b.unit_file]
    specials = []
    extra = []      # rows of files read as text (non-ASCII sequence characters)
    keep = None
    io_error = None
    bad = b.bad != 0 if b.bad is not None and len(b.bad) else np.zeros(b.n_files, dtype=bool)
    if not b.regular or bad.any():
        st = b.status
        if (st == PFA_ERR_IO).any():
            # a file that cannot be read (a subdirectory inside --dir, a vanished or unreadable file): the reference prints every
            # row before it and dies with the OSError at its open() (PolyFastA.py:104-108, :239); the caller does the same
            # (run_files: rows of the files below it, flush, then raise_unreadable).  Such files yield no rows here.
            io_error = int(b.file_idx[st == PFA_ERR_IO].min())      # (a file that is not text may join it below)
        bad = bad | (st == PFA_FILE_UNSUPPORTED)      # unequal BYTE lengths and non-ASCII characters: the loader could not tell
        odd = (st != 0) | bad
        if args.cds:
            odd |= (st == 0) & (b.n_sites % 3 != 0)
        if b.key_rows is not None:
            odd |= (st == 0) & (b.key_rows == 0).any(axis=1)
        for k in np.flatnonzero(odd).tolist():
            fidx = int(b.file_idx[k])
            path = paths[fidx]
            name = path.split("/")[-1]
            if st[k] == PFA_ERR_IO:
                continue
            if st[k] == PFA_FILE_NOT_FASTA:
                specials.append((fidx, 0, "stdout", f"# file {path} is not FASTA!"))                        # :247 (always stdout)
            elif st[k] == PFA_FILE_UNEQUAL:
                specials.append((fidx, 0, "comment", f"# Sequences do not have the same length: {name}"))   # :135-140
            elif bad[k] or (b.key_rows is not None and not b.key_rows[k].any() and not _file_is_ascii(path)):
                # bytes that are not ASCII: this file is read as text, the way the reference reads it.  It is the card's repack that
                # tells them; of a file in which NO population key was found no row went to the card, so the host looks itself (one
                # pass over its bytes) — what such a file prints (its length modulo 3, whether its sequences are equally long in
                # CHARACTERS) is the text reader's call (found by tools/g14_explore.py: three characters in four bytes)
                got = None
                try:
                    with open(path, "r") as fh:                                                              # :239
                        got = text_alignment_units(fasta.read_text_alignment(fh), path, args, keys)
                except UnicodeDecodeError:
                    # not text in the locale's encoding: the reference dies in its read loop (:239-240) after the rows of the
                    # files before this one — the caller's way out for a file that cannot be read (run_files: raise_unreadable)
                    io_error = fidx if io_error is None else min(io_error, fidx)
                    continue
                if got is None:
                    specials.append((fidx, 0, "comment", f"# Sequences hold non-ASCII characters (not supported, one byte per base): {name}"))
                else:
                    extra.extend((fidx,) + u for u in got[0])
                    specials.extend((fidx,) + x for x in got[1])
            else:
                if args.cds and int(b.n_sites[k]) % 3 != 0:                                                  # :114-119
                    specials.append((fidx, 0, "comment", f"# CDS sequence length is not a multiple of 3: {name}"))
                if b.key_rows is not None:
                    for j in np.flatnonzero(b.key_rows[k] == 0).tolist():                                    # :126-131
                        specials.append((fidx, 1 + j, "comment", f"# Pop {keys[j]} string was not found in fasta headers."))
        if bad.any():
            keep = ~bad[b.unit_file]
    ssx = None
    if syn is not None:
        ssx = b.sites_s if b.sites_s is not None else np.full(nu, np.nan, dtype=np.float64)
    sfs = None
    if args.sfs and nu:
        off = b.sfs_offsets().tolist()
        nbad = res["sfs_bad"].tolist()
        nr = b.unit_rows.tolist()
        sfs = [None if nbad[i] else sfs_flat[off[i]:off[i] + nr[i] // 2].tolist() for i in range(nu)]
    if keep is not None:
        meta, res = meta[keep], res[keep]
        syn = None if syn is None else syn[keep]
        ssx = None if ssx is None else ssx[keep]
        sfs = None if sfs is None else [x for x, k in zip(sfs, keep.tolist()) if k]
    if extra:
        from ._lib import PFA_SYN_BINS
        em = np.zeros(len(extra), dtype=META_DTYPE)
        em["fidx"], em["key"] = [u[0] for u in extra], [u[1] for u in extra]
        em["n_rows"], em["n_sites"] = [u[2] for u in extra], [u[3] for u in extra]
        meta = np.concatenate([meta, em])
        res = np.concatenate([res, np.array([u[4] for u in extra], dtype=res.dtype)])
        if args.cds:
            more = np.array([u[5] for u in extra], dtype=np.uint32).reshape(-1, PFA_SYN_BINS)
            syn = more if syn is None else np.concatenate([syn, more])
            more_x = np.array([np.nan if u[7] is None else u[7] for u in extra], dtype=np.float64)
            ssx = more_x if ssx is None else np.concatenate([ssx, more_x])
        if args.sfs:
            sfs = (sfs or []) + [u[6] for u in extra]
    return RowBlock(meta, res, syn, sfs, specials, io_error, ssx=ssx)


def _file_is_ascii(path):
    try:
        with open(path, "rb") as fh:
            while True:
                chunk = fh.read(1 << 22)
                if not chunk:
                    return True
                if not chunk.isascii():
                    return False
    except OSError:
        return True      # (cannot be read: the caller's other branches deal with it)


def raise_unreadable(path):
    """the error the reference dies with reading this path (PolyFastA.py:239-240): the OSError of its open() —
    IsADirectoryError, FileNotFoundError, PermissionError ... — or the UnicodeDecodeError of its read loop over a file that
    is not text in the locale's encoding"""
    with open(path, "r") as f:
        for _ in f:
            pass
    raise OSError(f"cannot read {path}")


def block_events(blk, paths, keys):
    """The block as the ordered event list `emit()` takes (the per-row path: --sfs, special
    lines, rows the native formatter hands back)."""
    from ._lib import PFA_NO_KEY
    items = []
    m = blk.meta.tolist()
    rec = blk.rec.tolist()
    for i, (fidx, key, n_rows, n_sites) in enumerate(m):
        pop = "NA" if key == PFA_NO_KEY else keys[key]
        items.append((fidx, 0 if key == PFA_NO_KEY else 1 + key, 1,
                      (fidx, "unit", paths[fidx].split("/")[-1], pop, n_rows, n_sites, rec[i],
                       None if blk.syn is None else blk.syn[i], None if blk.sfs is None else blk.sfs[i],
                       None if blk.ssx is None else float(blk.ssx[i]))))
    for fidx, sub, kind, text in blk.specials:
        items.append((fidx, sub, 0, (fidx, kind, text)))
    items.sort(key=lambda t: t[:3])
    return [t[3] for t in items]


def format_block_native(blk, table, coding, jc):
    """Every row of a block in one call into the library (csrc/pfa_format.cpp:
    pfa_format_rows_indexed); names and population keys are looked up natively in `table`
    (ingest.NameTable).  None when some row needs the Python path (the reference would raise there)."""
    import ctypes
    from . import _lib
    n = len(blk.meta)
    m = blk.meta
    fidx, key, nrows, nsites = (np.ascontiguousarray(m[f]) for f in ("fidx", "key", "n_rows", "n_sites"))
    a1, e1, e2 = _sample_size_constants(nrows)
    rec = np.ascontiguousarray(blk.rec)
    syn = np.ascontiguousarray(blk.syn) if coding else None
    ssx = np.ascontiguousarray(blk.ssx, dtype=np.float64) if coding and blk.ssx is not None else None
    out = np.empty((448 if coding else 192) * n + 512 + table.row_bytes(fidx), dtype=np.uint8)
    bad = ctypes.c_int64(-1)
    got = _lib.load().pfa_format_rows_indexed(_lib.ptr(table.names_blob), _lib.ptr(table.names_off), _lib.ptr(table.keys_blob),
                                              _lib.ptr(table.keys_off), n, _lib.ptr(fidx), _lib.ptr(key), _lib.ptr(nrows),
                                              _lib.ptr(nsites), _lib.ptr(rec), _lib.ptr(syn), _lib.ptr(ssx), _lib.ptr(a1), _lib.ptr(e1), _lib.ptr(e2),
                                              int(bool(jc)), int(bool(coding)), _lib.ptr(out), len(out), ctypes.byref(bad))
    if got < 0:
        if bad.value >= 0:
            return None
        _lib.check(got)
    return out[:got].tobytes().decode("utf-8", "surrogateescape")


def emit_block(blk, sink, args, table, paths, keys):
    """Write a block: natively formatted in one piece while it holds nothing but rows, else event by event."""
    from ._lib import PFA_NO_KEY
    n = len(blk.meta)
    if n == 0 and not blk.specials:
        return
    if n and not blk.specials and not args.sfs:
        text = format_block_native(blk, table, args.cds, args.jc)
        if text is not None:
            if sink.bulk_ok():
                sink.rows(text)
            else:    # --out without --silent: the reference's progress chatter, one piece per row, then the rows
                names = [paths[i].split("/")[-1] for i in blk.meta["fidx"].tolist()]
                pops = ["NA" if k == PFA_NO_KEY else keys[k] for k in blk.meta["key"].tolist()]
                sink.rows_with_progress(text, names, pops)
            return
    emit(block_events(blk, paths, keys), sink, args.cds, args.jc, args.sfs)


startup_marks = []  # (phase that ENDS here, wall clock): __main__ and main() append; printed when PFA_TRACE_STARTUP is set
last_stats = {}     # what the last run_files() measured (bench.py `end_to_end`: the stage breakdown)


def text_runs(blk, table, args):
    """The rows of a sorted block as text, one piece per run of consecutive files: [(first file index, text)], or None
    when some row needs the Python path.  (A rank of a multi-GPU job owns whole claims of the shared queue: its rows are
    a few dozen such runs.)"""
    fidx = blk.meta["fidx"].astype(np.int64)
    if len(fidx) == 0:
        return []
    cuts = np.flatnonzero(np.diff(fidx) > 1) + 1
    out = []
    for a, b in zip(np.concatenate(([0], cuts)).tolist(), np.concatenate((cuts, [len(fidx)])).tolist()):
        part = RowBlock(blk.meta[a:b], blk.rec[a:b], None if blk.syn is None else blk.syn[a:b], ssx=None if blk.ssx is None else blk.ssx[a:b])
        text = format_block_native(part, table, args.cds, args.jc)
        if text is None:
            return None
        out.append((int(fidx[a]), text))
    return out


def run_files(paths, args, sink, rank, world, comm, keys, directory=None):
    """`paths`: the sorted file list — or None with `directory`: the library lists and sorts it itself
    (PolyFastA.py:93-94, :104; falls back to os.listdir where it declines).
    The file loop (PolyFastA.py:104-140) on the streaming ingest.  One rank: rows leave batch by
    batch while later files are still being read.  Several ranks: every rank drains the shared
    queue, the blocks travel to rank 0 once (`comm.gather_bytes`), rank 0 orders and writes them."""
    import time
    from . import ingest
    trace = [] if os.environ.get("PFA_TRACE") else None      # per-batch timeline on stderr (tools/e2e_modes_probe.py)
    t00 = time.perf_counter()
    ing = None
    if paths is None:
        ing = (ingest.Ingest.from_dir_shared(directory, comm, keys=keys, cds=args.cds) if world > 1 and comm is not None
               else ingest.Ingest.from_dir(directory, keys=keys, cds=args.cds))
        if ing is None:                      # a name that is not UTF-8, or a directory that cannot be listed: the reference's own way
            paths = sorted(directory + "/" + i for i in os.listdir(directory))
        else:
            paths = ing.paths
    if ing is None:
        ing = ingest.Ingest(paths, keys=keys, cds=args.cds)
    # the ranks of a node share the file loop through a work queue (claim order = path order, the same on every rank)
    from . import dist
    wq = dist.WorkQueue(len(paths), world) if world > 1 else None
    t_open = time.perf_counter() - t00
    tm = {"next_wait_s": 0.0, "block_s": 0.0, "format_s": 0.0, "flags_gather_s": 0.0, "gather_s": 0.0, "gather_bytes": 0,
          "root_merge_write_s": 0.0, "barrier_s": 0.0}      # stage timers of this rank (bench.py: end_to_end_ranks gathers them)
    if trace is not None:
        trace.append(("open", time.perf_counter() - t00, len(paths)))
    try:
        drained = [wq is None]
        if wq is None:
            ing.feed(None, last=True)

        def top_up():
            # keep two claims' worth of files in the loader's hands; one claim = one grain of the queue
            while not drained[0] and ing.backlog() < 2 * wq.grain:
                got = wq.claim()
                if not got:
                    drained[0] = True
                ing.feed(got, last=not got)
        kept = []
        # one rank prints as it goes, in file order: batches hold SETS of files (the loaders place them as they
        # finish), so rows wait in `kept` until every file before them has arrived
        arrived = np.zeros(len(paths) + 1, dtype=bool)
        upto = 0
        io_fail = None       # index of the first file that could not be read: rows of the files before it, then the reference's OSError
        t_fetch = t_emit = 0.0
        while True:
            top_up()
            t0 = time.perf_counter()
            b = ing.next()
            if b is None:
                break
            t1 = time.perf_counter()
            res, syn, sfs_flat = b.fetch(args.cds, want_sfs=args.sfs)
            t2 = time.perf_counter()
            blk = block_from_batch(b, res, syn, sfs_flat, args, paths, keys)
            kept.append(blk)
            if blk.io_error is not None:
                io_fail = blk.io_error if io_fail is None else min(io_fail, blk.io_error)
            if world == 1:
                arrived[b.file_idx] = True
                new_upto = upto + int(np.argmin(arrived[upto:]))        # first file that has not arrived yet
                if new_upto > upto:
                    upto = new_upto
                    allb = RowBlock.concat(kept, args.cds).sorted()
                    ready = allb.split_below(upto if io_fail is None else min(upto, io_fail))
                    kept = [ready[1]]
                    emit_block(ready[0], sink, args, ing.table, paths, keys)
                if io_fail is not None and upto >= io_fail:            # every file before the unreadable one has printed its rows
                    sink.flush()
                    sys.stdout.flush()
                    raise_unreadable(paths[io_fail])
            t3 = time.perf_counter()
            tm["next_wait_s"] += t1 - t0
            t_fetch += t2 - t1
            t_emit += t3 - t2
            if trace is not None:
                trace.append(("batch", t0 - t00, b.n_files, t1 - t0, t2 - t1, t3 - t2))
        if world == 1 and io_fail is not None:
            raise_unreadable(paths[io_fail])
        fail_at = None       # several ranks: the first unreadable file of ALL ranks (every rank takes part in the exchange below)
        if world > 1:
            # Every rank formats its own rows (natively, in parallel) as runs of consecutive files — a claim of the queue is
            # such a run — and rank 0 only puts the runs of all ranks in file order: its share of the work does not grow with
            # the number of ranks.  When some rank holds anything but plain rows (comment lines, --sfs, the progress chatter of
            # --out without --silent, a row the native formatter hands back) the blocks travel instead and rank 0 formats.
            import pickle
            tg0 = time.perf_counter()
            mine = RowBlock.concat(kept, args.cds).sorted()
            runs = text_runs(mine, ing.table, args) if (not mine.specials and not args.sfs and sink.bulk_ok()) else None
            tg1 = time.perf_counter()
            # one exchange: "my rows are plain text runs" and the index of my first unreadable file (-1: none)
            mine_flags = np.array([int(runs is not None), -1 if io_fail is None else int(io_fail)], dtype=np.int64)
            flags = comm.all_gather_fixed(mine_flags.view(np.uint8)).view(np.int64).reshape(-1, 2)     # (also where a rank waits for the slowest one)
            tg2 = time.perf_counter()
            bad = flags[:, 1][flags[:, 1] >= 0]
            if len(bad):
                # The reference prints every row before the unreadable file and dies at its open() (PolyFastA.py:104-108, :239): the rows
                # below it travel to rank 0 as blocks, rank 0 writes them and raises the error; the other ranks leave with its exit code.
                fail_at = int(bad.min())
                mine = mine.split_below(fail_at)[0]
                runs = None
            if fail_at is None and bool(flags[:, 0].all()):
                blob = pickle.dumps(runs, protocol=pickle.HIGHEST_PROTOCOL)
                tg3 = time.perf_counter()
                parts = comm.gather_bytes(blob, dst=0)
                tg4 = time.perf_counter()
                if rank == 0:
                    allruns = sorted((r for p in parts for r in pickle.loads(p)), key=lambda r: r[0])
                    if allruns:
                        sink.rows("".join(r[1] for r in allruns))
            else:
                blob = mine.to_bytes()
                tg3 = time.perf_counter()
                parts = comm.gather_bytes(blob, dst=0)
                tg4 = time.perf_counter()
                if rank == 0:
                    allb = RowBlock.concat([RowBlock.from_bytes(p) for p in parts], args.cds).sorted()
                    emit_block(allb, sink, args, ing.table, paths, keys)
            tg5 = time.perf_counter()
            tm.update(format_s=(tg1 - tg0) + (tg3 - tg2), flags_gather_s=tg2 - tg1, gather_s=tg4 - tg3, gather_bytes=len(blob),
                      root_merge_write_s=tg5 - tg4)
        if trace is not None:
            trace.append(("loop_end", time.perf_counter() - t00))
        if fail_at is not None:
            sink.flush()
            sys.stdout.flush()
            if comm is not None:
                comm.barrier()
            # Rank 0 dies first and in full: a launcher (torchrun, dist.spawn_ranks) ends the other ranks as soon as one of them
            # leaves with an error, so the reference's traceback is on stderr before anybody exits; the others wait for it.
            if rank == 0:
                try:
                    raise_unreadable(paths[fail_at])
                except BaseException:      # noqa: BLE001
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
            if comm is not None:
                comm.barrier()
            raise SystemExit(1)
        tb = time.perf_counter()
        if comm is not None:
            comm.barrier()
        t4 = time.perf_counter()
        tm["barrier_s"] = t4 - tb
        sink.flush()
        st = ing.stats()
        st.update(files=len(paths), open_s=t_open, wall_s=time.perf_counter() - t00, fetch_wait_s=t_fetch, rows_s=t_emit, flush_s=time.perf_counter() - t4)
        st.update(tm)
        if wq is not None:
            st.update(claims=wq.n_claims, claim_s=wq.claim_s, claimed_files=wq.n_items, queue_open_s=wq.open_s)
        last_stats.clear()
        last_stats.update(st)
        if trace is not None:
            print("PFA_TRACE", {k: (round(v, 5) if isinstance(v, float) else v) for k, v in st.items()}, "total", round(time.perf_counter() - t00, 5), file=sys.stderr)
            for t in trace:
                print("PFA_TRACE", t[0], *(round(x, 5) if isinstance(x, float) else x for x in t[1:]), file=sys.stderr)
        return st
    finally:
        t9 = time.perf_counter()
        if wq is not None:
            wq.close(unlink=rank == 0)       # (success: every rank is past its last claim, the barrier above; failure: the others keep their descriptor)
        ing.close()
        last_stats["close_s"] = time.perf_counter() - t9          # (queue descriptor, loader threads joined, buffers back in the cache)
        if trace is not None:
            print("PFA_TRACE close", round(time.perf_counter() - t9, 5), file=sys.stderr)


def main(argv=None, comm=None, close_comm=True):
    """`comm`: a dist.Comm to use instead of making one (tests inject gloo; bench.py lends its own and
    keeps it: `close_comm=False`); with WORLD_SIZE > 1 and no `comm` the ranks talk through dist.RcclComm."""
    import time
    t_main = time.perf_counter()
    mark = (lambda name: startup_marks.append((name, time.time()))) if os.environ.get("PFA_TRACE_STARTUP") else (lambda name: None)
    parser = build_parser()
    args = parser.parse_args(argv)
    mark("argparse")

    if args.file == "" and args.dir == "." and not args.pipe:                      # PolyFastA.py:84-88
        print("Both --file/-f and --dir/-d were not found.")
        r = input("Do you wish to run polySFS on all files in the current directory? [y|n]: ")
        if r != "y":
            parser.error(parser.print_help())
    if len(args.file) != 0 and args.dir != ".":                                    # :90-91
        parser.error("Run with either --file/-f or --dir/-d, but not both")
    from . import _lib, dist
    if comm is None and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # one process per GPU: start them now, before anything here has touched the GPU or the files
        return dist.spawn_ranks(args.gpus, [sys.executable, "-m", "polyfasta_amd"] + list(sys.argv[1:] if argv is None else argv))
    dir_mode = len(args.file) == 0 and args.dir != "."                             # :93-94: every entry of the directory,
    if dir_mode and args.pipe:                                                     #   (with --pipe the listing only serves the error of :105-106)
        args.file = [args.dir + "/" + i for i in os.listdir(args.dir)]
        dir_mode = False
    if dir_mode:                                                                   #   listed and sorted by the library (run_files)
        pass
    elif len(args.file) == 0 and args.dir == "." and args.pipe:                    # :96-100
        args.file = ["stdin"] if not args.name else [args.name]
    if len(args.file) > 0 and args.pipe and not (args.file[0] == "stdin" or args.file[0] == args.name):
        parser.error("A FASTA file or multiple files cannot be used with the --pipe argument.")   # :105-106

    rank, _, world = dist.env_world()
    if comm is not None:
        rank, world = comm.rank, comm.world
    elif args.gpus > 1 and world != args.gpus:
        print(f"polyfasta_amd: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        return 2
    if comm is None and world > 1 and not dist.single_node(world):
        print("polyfasta_amd: the ranks share a work queue and a rendezvous file of ONE node; "
              f"LOCAL_WORLD_SIZE={os.environ.get('LOCAL_WORLD_SIZE')} != WORLD_SIZE={world} (run one job per node)", file=sys.stderr)
        return 2
    # the CPUs next to this rank's GPU, before the first GPU call: pinned buffers and loader threads then
    # sit on the socket the card hangs off (H2D from the other socket ran 33 instead of 54 GB/s here)
    dist.bind_to_gpu_node(dist.env_world()[1], int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    try:                                   # the scan runs on the GPU or not at all: say so before any output
        _lib.load()
        mark("dlopen libpfa_hip.so (+ libamdhip64)")
        _lib.init()                        # GPU LOCAL_RANK
        mark("pfa_init (hipInit, device, stream)")
        if comm is None and (world > 1 or os.environ.get("PFA_FORCE_DIST")):   # PFA_FORCE_DIST: one-rank RCCL rehearsal
            comm = dist.RcclComm()
    except (_lib.PfaError, _lib.PfaLibraryError, TimeoutError) as e:
        print(f"polyfasta_amd: {e}", file=sys.stderr)
        return 2

    sink = Sink(args.out, args.silent)
    if rank == 0:
        sink.header(args.cds, args.sfs)                                            # :102-103
    paths = None if dir_mode and not args.pipe else sorted(args.file)              # :104
    keys = args.pops.split(",") if args.pops else []                               # :123
    if args.pipe:
        events = process_pipe([(0, paths[0])] if rank == 0 and paths else [], args)
        if comm is not None:
            comm.barrier()
        if rank == 0:
            emit(events, sink, args.cds, args.jc, args.sfs)
    else:
        if os.environ.get("PFA_TRACE"):
            print("PFA_TRACE before_run_files", round(time.perf_counter() - t_main, 5), file=sys.stderr)
        t_pre = time.perf_counter() - t_main
        run_files(paths, args, sink, rank, world, comm, keys, directory=args.dir if dir_mode else None)
        last_stats["main_pre_s"] = t_pre                  # argparse, affinity, library, header (before the file loop)
        if os.environ.get("PFA_TRACE"):
            print("PFA_TRACE after_run_files", round(time.perf_counter() - t_main, 5), file=sys.stderr)
    mark("files -> rows (ingest, first kernel launches, fetch, print)")
    if rank == 0 and len(args.out) != 0 and not args.silent:                       # :141-142
        print("")
    if comm is not None and close_comm:
        comm.close()
    if os.environ.get("PFA_TRACE_STARTUP"):
        import json
        print("PFA_STARTUP " + json.dumps(startup_marks), file=sys.stderr)
    last_stats["main_s"] = time.perf_counter() - t_main
    return 0


if __name__ == "__main__":
    sys.exit(main())
