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
from ._lib import (DESC_DTYPE, PFA_ERR_IO, PFA_FILE_DEFERRED, PFA_FILE_NOT_FASTA, PFA_FILE_UNEQUAL,
                   PFA_FILE_UNSUPPORTED, PFA_FLAG_CDS, PFA_NO_ROW_IDX)
from .scan import Batch, DeviceBuffer, Plan, finalize_layout, scan_batch

HEADER_PLAIN = "file,seqlen,pop,N,seg_sites,pi,theta,tajimasD"
HEADER_CDS = ("file,sites_S,sites_N,pop,N,seg_sites_S,seg_sites_N,pi_S,pi_N,theta_S,theta_N,"
              "tajimasD_S,tajimasD_N,nstops")
NO_VAR = "0,0,0,NA"

# two pinned ingest buffers (allocated once per process): the loader threads fill one while
# the other is uploaded and scanned
BATCH_BYTES = 384 << 20
FILES_PER_CALL = 8192
_pinned_cache = [None, None]


def _ingest_buffer(slot, min_bytes=0):
    """Process-wide pinned ingest buffers (hipHostMalloc costs ~60 ms per GB, so they
    are sized once and reused by every batch)."""
    want = max(BATCH_BYTES, int(min_bytes))
    if _pinned_cache[slot] is None or len(_pinned_cache[slot].array) < want:
        if _pinned_cache[slot] is not None:
            _pinned_cache[slot].free()
        _pinned_cache[slot] = fasta.PinnedBuffer(want)
    return _pinned_cache[slot]


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


def format_cds_row(name, pop, n_rows, seqlen, res, syn_num, jc):
    """One coding CSV row (PolyFastA.py:151-180), including the reference's
    mis-ordered no-variation row (:160-162)."""
    r = _fields(res)
    ssites = stats.syn_sites_from_bins(syn_num)
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


_device_cache = [None]


def _device_buffer(nbytes):
    """One device buffer for the ingest batches of this process (a batch is scanned and its
    records fetched before the next one is uploaded), grown on demand."""
    if _device_cache[0] is None or _device_cache[0].nbytes < nbytes:
        if _device_cache[0] is not None:
            _device_cache[0].free()
        _device_cache[0] = DeviceBuffer(max(int(nbytes), 16))
    return _device_cache[0]


def _scan_uploaded(dev, used, descs, row_idx):
    """Scan every unit of a batch whose bases are already on their way to `dev`."""
    table = finalize_layout(descs, row_idx, used)
    plan = Plan(table)
    try:
        plan.run(dev)
        return plan.fetch()          # synchronises the stream: upload, scan and record copy are done
    finally:
        plan.close()


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
            done.append((fidx, "unit", name, pop, n_rows, seqlen, recs[unit], syn, sfs))
        else:
            done.append(ev)


def process_files(indexed_paths, args, more=None):
    """Ingest (native multi-threaded packer), scan (one launch per batch) and
    return the ordered output events of the given (index, path) pairs
    (PolyFastA.py:104-140).  `more()` (multi-GPU work queue) hands out further pairs
    whenever the local list runs dry, [] when there are none left."""
    done = []
    if args.pipe:
        for fidx, path in indexed_paths:
            from ._lib import PFA_ERR_UNSUPPORTED, PfaError
            try:
                aln = fasta.read_fasta_stdin(stride_align=128)
                status = PFA_FILE_NOT_FASTA if aln is None else (0 if aln.same_length else PFA_FILE_UNEQUAL)
            except PfaError as e:
                if e.code != PFA_ERR_UNSUPPORTED:
                    raise
                aln, status = None, PFA_FILE_UNSUPPORTED
            batch = Batch(stride_align=128)
            mid = batch.add_matrix(aln.matrix) if status == 0 else None
            pending = _events_for(fidx, path, path.split("/")[-1], status, aln.names if aln else None,
                                  aln.n_rows if aln else 0, int(aln.seq_lens[0]) if status == 0 else 0, args,
                                  lambda rows: batch.add_unit(mid, rows=None if rows is None else np.asarray(rows, dtype=np.uint32), cds=args.cds))
            _attach(pending, scan_batch(batch) if batch.n_units else None, done, args.sfs)
        return done

    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    queue = deque(indexed_paths)
    want_names = bool(args.pops)

    def refill():
        if more is not None and not queue:
            queue.extend(more())
        return bool(queue)

    def next_chunk(cap):
        """files whose text adds up to at most ~85 % of a buffer (packed size ~ text size)"""
        chunk, size = [], 0
        # (shared work queue: one claim per batch — a rank never hoards what another could scan meanwhile)
        while (queue or (not chunk and refill())) and len(chunk) < FILES_PER_CALL:
            try:
                sz = os.path.getsize(queue[0][1])
            except OSError:
                sz = 0
            if chunk and size + sz > 0.85 * cap:
                break
            chunk.append(queue.popleft())
            size += sz
        return chunk

    def load(chunk, slot, min_bytes=0):
        buf = _ingest_buffer(slot, min_bytes)
        return chunk, buf, fasta.load_files([p for _, p in chunk], buf.array, stride_align=128, want_names=want_names)

    pool = ThreadPoolExecutor(max_workers=1)     # the native loader runs its own thread pool; this thread only overlaps it
    try:
        slot = 0
        first = next_chunk(BATCH_BYTES)
        fut = pool.submit(load, first, slot) if first else None
        while fut is not None:
            chunk, pinned, (info, names, used) = fut.result()
            status = info["status"]
            if (status == PFA_ERR_IO).any():
                path = chunk[int(np.flatnonzero(status == PFA_ERR_IO)[0])][1]
                open(path, "rb").close()          # raises the OSError the reference would raise (:239)
                raise OSError(f"cannot read {path}")
            dev = None
            if used and (status == 0).any():
                # the H2D copy starts now and runs under the host work below (next ingest request,
                # descriptor table, plan); the pinned buffer is not touched again before the fetch
                dev = _device_buffer(used)
                dev.upload_async(pinned.array[:used])
            nxt = next_chunk(BATCH_BYTES)
            fut = pool.submit(load, nxt, 1 - slot) if nxt else None      # ingest of the next batch overlaps this one's scan
            deferred = [chunk[k] for k in np.flatnonzero(status == PFA_FILE_DEFERRED)]
            descs, ridx, pending = [], None, []
            if not args.pops:
                # one unit per packed file: build the descriptor table without a Python loop
                ok = np.flatnonzero(status == 0)
                descs = np.zeros(len(ok), dtype=DESC_DTYPE)
                for f in ("base_off", "n_sites", "row_stride"):
                    descs[f] = info[f][ok]
                descs["n_rows"] = descs["mat_rows"] = info["n_rows"][ok]
                descs["row_idx_off"] = PFA_NO_ROW_IDX
                descs["flags"] = PFA_FLAG_CDS if args.cds else 0
                unit_of = dict(zip(ok.tolist(), range(len(ok))))
                for k, fi in enumerate(info.tolist()):
                    st = fi[0]
                    if st == PFA_FILE_DEFERRED:
                        continue
                    fidx, path = chunk[k]
                    pending += _events_for(fidx, path, path.split("/")[-1], st, None, fi[1], fi[2], args,
                                           lambda rows, k=k: unit_of[k])
            else:
                # one unit per (file, population): descriptor columns collected as Python lists, the table
                # built once per batch
                cols = {f: [] for f in ("base_off", "n_sites", "row_stride", "mat_rows", "n_rows", "row_idx_off")}
                rl, nri = [], [0]
                for k, fi in enumerate(info.tolist()):
                    st = fi[0]
                    if st == PFA_FILE_DEFERRED:
                        continue
                    fidx, path = chunk[k]

                    def add_unit(rows, fi=fi):
                        cols["base_off"].append(fi[4])
                        cols["n_sites"].append(fi[2])
                        cols["row_stride"].append(fi[3])
                        cols["mat_rows"].append(fi[1])
                        cols["n_rows"].append(len(rows))
                        cols["row_idx_off"].append(nri[0])
                        nri[0] += len(rows)
                        rl.append(rows)
                        return len(rl) - 1
                    pending += _events_for(fidx, path, path.split("/")[-1], st, names[k], fi[1], fi[2], args, add_unit)
                descs = np.zeros(len(rl), dtype=DESC_DTYPE)
                for f, v in cols.items():
                    descs[f] = v
                descs["flags"] = PFA_FLAG_CDS if args.cds else 0
                ridx = np.fromiter((r for rows in rl for r in rows), dtype=np.uint32, count=nri[0]) if rl else None
            if len(descs):
                out = _scan_uploaded(dev, used, descs, ridx)
            else:
                out = None
                if dev is not None:
                    from . import _lib
                    _lib.check(_lib.load().pfa_sync())
            _attach(pending, out, done, args.sfs)
            if deferred:
                if len(deferred) == len(chunk):
                    # nothing fitted: the first file alone is larger than the buffer.  Let the ingest in
                    # flight finish, then load that file alone into a buffer grown to its size.
                    if fut is not None:
                        queue.extendleft(reversed(fut.result()[0]))
                    need = (int(info[0]["row_stride"]) << 8) + (1 << 20)
                    queue.extendleft(reversed(deferred[1:]))
                    fut = pool.submit(load, deferred[:1], slot, need)
                    continue
                queue.extendleft(reversed(deferred))
                if fut is None:
                    nxt = next_chunk(BATCH_BYTES)
                    fut = pool.submit(load, nxt, 1 - slot) if nxt else None
            slot = 1 - slot
    finally:
        pool.shutdown(wait=True)
    done.sort(key=lambda ev: ev[0])       # stable: deferred files return to their sorted position
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
    (name, pop, n_rows, seqlen, record tuple, syn_num row).  Returns the text block, or None when
    some row needs the Python path."""
    import ctypes
    from . import _lib
    from ._lib import RESULT_DTYPE
    n = len(units)
    names, noff = _blob([f"{u[0]}," for u in units])
    mids, moff = _blob([f",{u[1]},{u[2]}," for u in units])
    rec = np.array([u[4] for u in units], dtype=RESULT_DTYPE)
    syn = np.ascontiguousarray(np.stack([np.asarray(u[5], dtype=np.uint32) for u in units]))
    nrows = np.fromiter((u[2] for u in units), dtype=np.uint32, count=n)
    seqlen = np.fromiter((u[3] for u in units), dtype=np.uint32, count=n)
    a1, e1, e2 = _sample_size_constants(nrows)
    out = np.empty(len(names) + len(mids) + 320 * n + 16, dtype=np.uint8)
    bad = ctypes.c_int64(-1)
    got = _lib.load().pfa_format_cds_rows(names, _lib.ptr(noff), mids, _lib.ptr(moff), n, _lib.ptr(rec), _lib.ptr(syn),
                                          _lib.ptr(nrows), _lib.ptr(seqlen), _lib.ptr(a1), _lib.ptr(e1), _lib.ptr(e2),
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
                    block = format_cds_rows_native([(e[2], e[3], e[4], e[5], _fields(e[6]), e[7]) for e in run], jc)
            if block is not None and sink.bulk_ok() and not sfs:
                sink.rows(block)
            else:
                rows = block.split("\n") if block is not None else None
                for k, e in enumerate(run):
                    name, pop, n_rows, seqlen, res, syn = e[2:8]
                    text = rows[k] if rows is not None else (
                        format_cds_row(name, pop, n_rows, seqlen, res, syn, jc) if cds
                        else format_plain_row(name, pop, n_rows, seqlen, res, jc))
                    if sfs:
                        text += "," + _sfs_text(e[8])
                    sink.row(text, name, pop)
            i = j
    sink.flush()


def gather_events(comm, events):
    """Multi-GPU: the integer records (and, for coding runs, the bins of synonymous-site
    numerators) travel as fixed-size structs through the communicator's all-gather (RCCL on
    the GPU box), the small text metadata (names, comment lines, --sfs vectors) as one pickled
    blob per rank.  Returns the merged, file-ordered event list on rank 0 (None elsewhere)."""
    from ._lib import PFA_SYN_BINS, RESULT_DTYPE
    units = [ev for ev in events if ev[1] == "unit"]
    rec = np.zeros(len(units), dtype=RESULT_DTYPE)
    if units:
        rec[:] = np.array([_fields(ev[6]) for ev in units], dtype=RESULT_DTYPE)
    any_syn = any(ev[7] is not None for ev in units)
    syn = np.zeros((len(units), PFA_SYN_BINS), dtype=np.uint32)
    for i, ev in enumerate(units):
        if ev[7] is not None:
            syn[i] = ev[7]
    all_rec = comm.all_gather_records(rec)
    all_syn = comm.all_gather_records(syn.reshape(-1) if any_syn else syn[:0].reshape(-1))
    meta = [ev[:6] + (ev[8],) if ev[1] == "unit" else ev for ev in events]      # no numpy payload
    all_meta = comm.gather_objects(meta, dst=0)
    if comm.rank != 0:
        return None
    merged = []
    for r in range(comm.world):
        k = 0
        syn_r = all_syn[r].reshape(-1, PFA_SYN_BINS) if len(all_syn[r]) else None
        rec_r = all_rec[r].tolist()
        for seq, ev in enumerate(all_meta[r]):
            if ev[1] == "unit":
                merged.append((ev[0], r, seq, ev[:6] + (rec_r[k], syn_r[k] if syn_r is not None else None, ev[6])))
                k += 1
            else:
                merged.append((ev[0], r, seq, ev))
    merged.sort(key=lambda t: (t[0], t[1], t[2]))
    return [t[3] for t in merged]


def _file_sizes(paths):
    sizes = []
    for p in paths:
        try:
            sizes.append(os.path.getsize(p))
        except OSError:
            sizes.append(0)
    return sizes


def main(argv=None, comm=None):
    """`comm`: a dist.Comm to use instead of RCCL (tests); with WORLD_SIZE > 1 and no `comm`
    the ranks talk through dist.RcclComm."""
    parser = build_parser()
    args = parser.parse_args(argv)

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
    if len(args.file) == 0 and args.dir != ".":                                    # :93-94
        args.file = [args.dir + "/" + i for i in os.listdir(args.dir)]
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
    try:                                   # the scan runs on the GPU or not at all: say so before any output
        _lib.init()                        # GPU LOCAL_RANK
        if comm is None and (world > 1 or os.environ.get("PFA_FORCE_DIST")):   # PFA_FORCE_DIST: one-rank RCCL rehearsal
            comm = dist.RcclComm()
    except (_lib.PfaError, _lib.PfaLibraryError, TimeoutError) as e:
        print(f"polyfasta_amd: {e}", file=sys.stderr)
        return 2

    sink = Sink(args.out, args.silent)
    if rank == 0:
        sink.header(args.cds, args.sfs)                                            # :102-103
    paths = sorted(args.file)                                                      # :104
    if args.pipe:
        mine = [(0, paths[0])] if rank == 0 and paths else []
    elif world > 1:
        mine = []          # everything comes out of the work queue the ranks share
    else:
        mine = list(enumerate(paths))
    wq = None
    if world > 1 and not args.pipe:
        wq = dist.WorkQueue(_file_sizes(paths), world)
    events = process_files(mine, args, more=(lambda: [(i, paths[i]) for i in wq.claim()]) if wq else None)
    if comm is not None:
        events = gather_events(comm, events)
        comm.barrier()
    if wq is not None:
        wq.close(unlink=rank == 0)       # every rank is past its last claim: the barrier above
    if rank == 0:
        emit(events, sink, args.cds, args.jc, args.sfs)
        if len(args.out) != 0 and not args.silent:                                 # :141-142
            print("")
    if comm is not None:
        comm.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
