"""Streaming ingest (csrc/pfa_ingest.hip through include/pfa.h: pfa_ingest_*): FASTA files ->
device-resident matrices -> launched scan plans, a batch at a time.

The feeder of the reference's file loop (PolyFastA.py:104-140: readfasta, the length check, the
population split).  Loader threads read every file once into pinned memory and index it; an
uploader thread keeps the PCIe link busy; `Ingest.next()` hands back whatever has reached the
card since the last call, already repacked into matrices and with its scan launched.  All of it
is native: this module only moves arrays.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from . import _lib
from ._lib import PFA_BATCH_REGULAR, PFA_SYN_BINS, RESULT_DTYPE


def _arr(ptr, n, dtype):
    """copy of `n` items behind a ctypes pointer (the ingest owns the memory until its next call)"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class IngestBatch:
    """A run of consecutive files of the feed order, scanned.  `file_idx` (indices into the
    ingest's paths), `status`, `n_rows`, `n_sites` per file; `key_rows[file, key]` = records whose
    name holds the key; per scan unit (= CSV row): `unit_file` (index into this batch), `unit_key`
    (PFA_NO_KEY without --pops), `unit_rows`."""

    def __init__(self, c, n_keys):
        nf, nu = int(c.n_files), int(c.n_units)
        self.n_files, self.n_units, self.first_pos = nf, nu, int(c.first_pos)
        self.regular = bool(c.flags & PFA_BATCH_REGULAR)
        self.file_idx = _arr(c.file_idx, nf, np.uint32)
        self.status = _arr(c.status, nf, np.int32)
        self.n_rows = _arr(c.n_rows, nf, np.uint32)
        self.n_sites = _arr(c.n_sites, nf, np.uint32)
        self.key_rows = _arr(c.key_rows, nf * n_keys, np.uint32).reshape(nf, n_keys) if n_keys else None
        self.unit_file = _arr(c.unit_file, nu, np.uint32)
        self.unit_key = _arr(c.unit_key, nu, np.uint32)
        self.unit_rows = _arr(c.unit_rows, nu, np.uint32)
        self._bad_ptr = c.bad_bytes
        self._plan = c.plan
        self.bad = None
        self.sites_s = None

    def fetch(self, coding, want_sfs=False):
        """Wait for the scan; -> (records RESULT_DTYPE[n_units], syn_num uint32[n_units, 65] or
        None, folded SFS (flat, unit i at sfs_off[i]) or None).  Also reads the per-file flags
        for bytes >= 0x80 in sequence lines (`self.bad`) and, for coding batches, re-scans the units whose
        synonymous-site count sits on a rounding tie while their matrices are still on the card
        (`self.sites_s`: NaN or the reference's own accumulation, scan.resolve_syn_ties)."""
        lib = _lib.load()
        nu = self.n_units
        res = np.zeros(nu, dtype=RESULT_DTYPE)
        syn = np.zeros((nu, PFA_SYN_BINS), dtype=np.uint32) if coding and nu else None
        sfs = None
        if self._plan:
            if want_sfs:
                sfs = np.zeros(max(int(lib.pfa_plan_sfs_words(self._plan)), 1), dtype=np.uint32)
            _lib.check(lib.pfa_plan_fetch(self._plan, _lib.ptr(res), _lib.ptr(sfs), _lib.ptr(syn), None, None, None))
        else:
            _lib.check(lib.pfa_sync())
        self.bad = _arr(self._bad_ptr, self.n_files, np.uint32)
        if syn is not None and self._plan:
            from .scan import resolve_syn_ties
            self.sites_s = resolve_syn_ties(self._plan, syn, self.n_sites[self.unit_file], missing_cols=res["missing_cols"])
        return res, syn, sfs

    def sfs_offsets(self):
        """offset of unit i's floor(N/2) bins in the flat SFS array"""
        bins = self.unit_rows.astype(np.int64) // 2
        return np.concatenate(([0], np.cumsum(bins)[:-1])) if len(bins) else np.zeros(0, dtype=np.int64)


class NameTable:
    """Display names (the text after the last '/', PolyFastA.py:110) of a path list and the
    population keys, as byte blobs with offsets: what pfa_format_rows_indexed looks rows up in."""

    def __init__(self, names_blob, names_off, keys):
        self.names_blob, self.names_off = names_blob, names_off
        self.keys = [k.encode("utf-8", "surrogateescape") if isinstance(k, str) else bytes(k) for k in keys]
        kb = b"".join(self.keys)
        self.keys_blob = np.frombuffer(kb, dtype=np.uint8).copy() if kb else np.zeros(1, dtype=np.uint8)
        self.keys_off = np.concatenate(([0], np.cumsum([len(k) for k in self.keys]))).astype(np.uint64)

    @staticmethod
    def from_paths(paths, keys=()):
        names = [os.fsencode(p).split(b"/")[-1] for p in paths]
        blob = b"".join(names)
        off = np.concatenate(([0], np.cumsum([len(x) for x in names]))).astype(np.uint64)
        return NameTable(np.frombuffer(blob, dtype=np.uint8).copy() if blob else np.zeros(1, dtype=np.uint8), off, keys)

    def row_bytes(self, fidx):
        """bytes the names and keys of these rows take (sizes an output buffer)"""
        klen = int(np.diff(self.keys_off).max()) if len(self.keys_off) > 1 else 2
        i = fidx.astype(np.int64)
        return int((self.names_off[i + 1] - self.names_off[i]).sum()) + (klen + 2) * len(fidx)


class NativePaths:
    """The path list of an ingest that listed its directory itself (Ingest.from_dir): a sequence of str,
    decoded on demand — a regular run never asks for one."""

    def __init__(self, ing, n):
        self._ing, self._n = ing, n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self._n))]
        if i < 0:
            i += self._n
        if not 0 <= i < self._n:
            raise IndexError(i)
        return os.fsdecode(_lib.load().pfa_ingest_path(self._ing.handle, i))

    def __iter__(self):
        return (self[i] for i in range(self._n))


class Ingest:
    """`paths`: every file the job may read; `feed()` says which of them, in which order, this
    process reads (all of them by default)."""

    @classmethod
    def from_dir(cls, directory, keys=(), cds=False, threads=0, stride_align=128, slot_bytes=0, n_slots=0):
        """The reference's `--dir` listing (`[dir + "/" + i for i in os.listdir(dir)]`, sorted; PolyFastA.py:93-94, :104)
        done by the library: 100 000 names are listed, sorted and kept natively.  Returns None when the library leaves the
        directory to the caller (a name that is not UTF-8, or a directory it cannot open: os.listdir then raises
        the reference's own error)."""
        _lib.init()
        lib = _lib.load()
        self = cls.__new__(cls)
        self.keys = [k.encode("utf-8", "surrogateescape") if isinstance(k, str) else bytes(k) for k in keys]
        karr = (ctypes.c_char_p * max(len(self.keys), 1))(*self.keys)
        o = _lib.IngestOptsC(n_threads=int(threads), stride_align=int(stride_align), unit_flags=_lib.PFA_FLAG_CDS if cds else 0,
                             n_keys=len(self.keys), keys=karr, slot_bytes=int(slot_bytes), n_slots=int(n_slots), reserved=0)
        h, n = ctypes.c_void_p(), ctypes.c_uint32()
        rc = lib.pfa_ingest_open_dir(os.fsencode(directory), ctypes.byref(o), ctypes.byref(h), ctypes.byref(n))
        if rc in (_lib.PFA_ERR_UNSUPPORTED, _lib.PFA_ERR_IO):
            self.handle = None
            return None
        _lib.check(rc)
        self.handle = h
        self.cds = bool(cds)
        self.paths = NativePaths(self, int(n.value))
        self._bind_names(int(n.value))
        return self

    @classmethod
    def from_dir_shared(cls, directory, comm, keys=(), cds=False, **kw):
        """`from_dir` for the ranks of a multi-GPU job: rank 0 lists and sorts the directory, the others take its name table
        (one all-gather) — R processes walking one directory of 100 k entries at the same time serialise on the directory
        (every rank's listing took 26 ms with 4 ranks, 51 ms with 6, against 5 ms alone), and every rank is certain to
        number the files alike.  None on every rank when the library declines the directory (the caller's os.listdir then)."""
        import struct
        ing = cls.from_dir(directory, keys=keys, cds=cds, **kw) if comm.rank == 0 else None
        mine = b""
        if ing is not None:
            n = len(ing.paths)
            mine = struct.pack("<Q", n) + np.ascontiguousarray(ing.table.names_off[:n + 1]).tobytes() + np.ascontiguousarray(ing.table.names_blob).tobytes()
        blob = comm.all_gather_bytes(mine)[0]
        if comm.rank == 0 or not blob:
            return ing
        n = struct.unpack_from("<Q", blob)[0]
        off = np.frombuffer(blob, dtype=np.uint64, count=n + 1, offset=8)
        names = np.frombuffer(blob, dtype=np.uint8, offset=8 + 8 * (n + 1))
        _lib.init()
        lib = _lib.load()
        self = cls.__new__(cls)
        self.keys = [k.encode("utf-8", "surrogateescape") if isinstance(k, str) else bytes(k) for k in keys]
        karr = (ctypes.c_char_p * max(len(self.keys), 1))(*self.keys)
        o = _lib.IngestOptsC(n_threads=int(kw.get("threads", 0)), stride_align=int(kw.get("stride_align", 128)), unit_flags=_lib.PFA_FLAG_CDS if cds else 0,
                             n_keys=len(self.keys), keys=karr, slot_bytes=int(kw.get("slot_bytes", 0)), n_slots=int(kw.get("n_slots", 0)), reserved=0)
        h = ctypes.c_void_p()
        self.handle = None
        _lib.check(lib.pfa_ingest_open_names(os.fsencode(directory), _lib.ptr(np.ascontiguousarray(names)) if len(names) else None,
                                             _lib.ptr(np.ascontiguousarray(off)), int(n), ctypes.byref(o), ctypes.byref(h)))
        self.handle = h
        self.cds = bool(cds)
        self.paths = NativePaths(self, int(n))
        self._bind_names(int(n))
        return self

    def _bind_names(self, n):
        lib = _lib.load()
        blob, off = ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(lib.pfa_ingest_names(self.handle, ctypes.byref(blob), ctypes.byref(off)))
        # views of the ingest's own name table: valid until close()
        noff = np.ctypeslib.as_array(ctypes.cast(off, ctypes.POINTER(ctypes.c_uint64)), shape=(n + 1,))
        nb = int(noff[-1])
        nblob = np.ctypeslib.as_array(ctypes.cast(blob, ctypes.POINTER(ctypes.c_uint8)), shape=(nb,)) if nb else np.zeros(1, dtype=np.uint8)
        self.table = NameTable(nblob, noff, self.keys)

    def __init__(self, paths, keys=(), cds=False, threads=0, stride_align=128, slot_bytes=0, n_slots=0):
        _lib.init()
        lib = _lib.load()
        self.paths = list(paths)
        self.keys = [k.encode("utf-8", "surrogateescape") if isinstance(k, str) else bytes(k) for k in keys]
        n = len(self.paths)
        # char*[n] over ONE joined buffer (a ctypes array of n Python bytes objects costs ~1 us per path)
        blob = None
        if all(isinstance(p, str) for p in self.paths):
            blob = ("\0".join(self.paths) + "\0").encode("utf-8", "surrogateescape")      # = os.fsencode of every path, in one call
            lens = np.fromiter(map(len, self.paths), dtype=np.uint64, count=n)
            if len(blob) != int(lens.sum()) + n:                                           # non-ASCII paths: lengths in bytes
                blob = None
        if blob is None:
            enc = [os.fsencode(p) for p in self.paths]
            blob = b"\0".join(enc) + b"\0"
            lens = np.fromiter(map(len, enc), dtype=np.uint64, count=n)
        self._path_blob = ctypes.create_string_buffer(blob, len(blob))
        offs = np.zeros(max(n, 1), dtype=np.uint64)
        if n > 1:
            np.cumsum(lens[:-1] + np.uint64(1), out=offs[1:n])
        offs += np.uint64(ctypes.addressof(self._path_blob))
        arr = offs.ctypes.data_as(ctypes.POINTER(ctypes.c_char_p))
        karr = (ctypes.c_char_p * max(len(self.keys), 1))(*self.keys)
        o = _lib.IngestOptsC(n_threads=int(threads), stride_align=int(stride_align), unit_flags=_lib.PFA_FLAG_CDS if cds else 0,
                             n_keys=len(self.keys), keys=karr, slot_bytes=int(slot_bytes), n_slots=int(n_slots), reserved=0)
        h = ctypes.c_void_p()
        _lib.check(lib.pfa_ingest_open(arr, n, ctypes.byref(o), ctypes.byref(h)))
        self.handle = h
        self.cds = bool(cds)
        self._bind_names(n)

    def feed(self, idx=None, last=True):
        lib = _lib.load()
        if idx is None:
            _lib.check(lib.pfa_ingest_feed(self.handle, None, len(self.paths), int(last)))
        else:
            a = np.ascontiguousarray(idx, dtype=np.uint32)
            _lib.check(lib.pfa_ingest_feed(self.handle, _lib.ptr(a) if len(a) else None, len(a), int(last)))

    def backlog(self) -> int:
        return int(_lib.load().pfa_ingest_backlog(self.handle))

    def next(self):
        """the next batch (its scan already launched), or None when every fed file has been handed out"""
        c = _lib.IngestBatchC()
        rc = _lib.check(_lib.load().pfa_ingest_next(self.handle, ctypes.byref(c)))
        return IngestBatch(c, len(self.keys)) if rc == 1 else None

    def stats(self):
        buf = (ctypes.c_double * 8)()
        _lib.check(_lib.load().pfa_ingest_stats(self.handle, buf))
        return {"read_thread_s": buf[0], "index_thread_s": buf[1], "bytes_read": int(buf[2]), "copy_busy_s": buf[3],
                "bytes_copied": int(buf[4]), "loader_threads": int(buf[5]), "turn_wait_thread_s": buf[6], "batches": int(buf[7])}

    def close(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            _lib.load().pfa_ingest_close(self.handle)
        self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
