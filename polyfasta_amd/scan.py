"""Batch scanning on the GPU: pack alignments, build a descriptor table, run
the HIP scan kernels through the C ABI, hand back integer summaries.

Vocabulary: a *matrix* is one alignment (N rows x L sites, one upper-case
ASCII byte per base, row-major with a 16-byte-aligned row stride); a *unit* is
what one CSV row of the reference describes — a matrix, or one population's
row subset of it (PolyFastA.py:122-134).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from ._lib import DESC_DTYPE, RESULT_DTYPE, PFA_FLAG_CDS, PFA_NO_ROW_IDX, PFA_SYN_BINS


def round_up(x: int, a: int) -> int:
    return (x + a - 1) // a * a


def row_stride_for(n_sites: int, align: int = 16) -> int:
    """Row stride the kernels accept: a multiple of 16 and at least one 16-byte chunk."""
    return max(16, round_up(int(n_sites), max(16, align)))


class DeviceBuffer:
    """RAII wrapper of a device allocation made through the C ABI."""

    def __init__(self, nbytes: int):
        _lib.init()
        p = ctypes.c_void_p()
        _lib.check(_lib.load().pfa_dev_alloc(int(nbytes), ctypes.byref(p)))
        self.ptr = p
        self.nbytes = int(nbytes)

    def upload(self, arr: np.ndarray, offset: int = 0):
        a = np.ascontiguousarray(arr)
        assert offset + a.nbytes <= self.nbytes
        _lib.check(_lib.load().pfa_upload(ctypes.c_void_p(self.ptr.value + offset), _lib.ptr(a), a.nbytes))

    def upload_async(self, arr: np.ndarray, offset: int = 0):
        """Enqueue the copy and return; `arr` (pinned memory) must stay untouched until the
        next fetch or pfa_sync()."""
        assert arr.flags["C_CONTIGUOUS"] and offset + arr.nbytes <= self.nbytes
        _lib.check(_lib.load().pfa_upload_async(ctypes.c_void_p(self.ptr.value + offset), _lib.ptr(arr), arr.nbytes))

    def download(self, nbytes: int = None, offset: int = 0) -> np.ndarray:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        _lib.check(_lib.load().pfa_download(_lib.ptr(out), ctypes.c_void_p(self.ptr.value + offset), nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.ptr.value:
            _lib.load().pfa_dev_free(self.ptr)
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


@dataclass
class UnitTable:
    """Descriptor table of a batch (host side)."""
    descs: np.ndarray                  # DESC_DTYPE[n_units]
    row_idx: np.ndarray                # uint32[]
    n_bases_bytes: int
    n_max: int = 0
    tile_sites: int = 256
    var_words: int = 4
    n_tiles: np.ndarray = field(default=None)

    @property
    def n_units(self) -> int:
        return len(self.descs)


def finalize_layout(descs: np.ndarray, row_idx, n_bases_bytes: int) -> UnitTable:
    """Fill sfs_off / tile_off (prefix sums in the geometry the library will use)."""
    lib = _lib.load()
    descs = np.ascontiguousarray(descs, dtype=DESC_DTYPE)
    n_max = int(descs["n_rows"].max()) if len(descs) else 0
    any_cds = int(bool(len(descs)) and bool(((descs["flags"] & PFA_FLAG_CDS) != 0).any()))
    tw = int(lib.pfa_tile_sites(n_max, any_cds))
    width = np.int64(tw)
    live = (descs["n_rows"] > 0) & (descs["n_sites"] > 0)
    n_tiles = np.where(live, (descs["n_sites"].astype(np.int64) + width - 1) // width, 0)
    bins = descs["n_rows"].astype(np.int64) // 2
    descs["sfs_off"] = np.concatenate(([0], np.cumsum(bins)[:-1])) if len(descs) else 0
    descs["tile_off"] = np.concatenate(([0], np.cumsum(n_tiles)[:-1])) if len(descs) else 0
    ri = np.ascontiguousarray(row_idx if row_idx is not None else np.zeros(0), dtype=np.uint32)
    return UnitTable(descs=descs, row_idx=ri, n_bases_bytes=int(n_bases_bytes), n_max=n_max,
                     tile_sites=tw, var_words=int(lib.pfa_var_words_per_tile(n_max, any_cds)), n_tiles=n_tiles)


class Batch:
    """Accumulates matrices and units on the host, then packs them into one
    contiguous base buffer + descriptor table."""

    def __init__(self, stride_align: int = 16):
        self.stride_align = stride_align
        self._mats = []      # (matrix uint8 [N, L], stride)
        self._mat_off = []
        self._units = []     # (mat_id, rows or None, cds)
        self._nbytes = 0

    def add_matrix(self, matrix: np.ndarray) -> int:
        m = np.asarray(matrix, dtype=np.uint8)
        if m.ndim != 2:
            raise ValueError("alignment matrix must be 2-D [rows, sites]")
        stride = row_stride_for(m.shape[1], self.stride_align)
        self._mats.append((m, stride))
        self._mat_off.append(self._nbytes)
        self._nbytes += round_up(max(m.shape[0], 1) * stride, 256)
        return len(self._mats) - 1

    def add_unit(self, mat_id: int, rows=None, cds: bool = False) -> int:
        if rows is not None:
            rows = np.asarray(rows, dtype=np.uint32)
        self._units.append((mat_id, rows, bool(cds)))
        return len(self._units) - 1

    @property
    def n_units(self) -> int:
        return len(self._units)

    @property
    def nbytes(self) -> int:
        return self._nbytes

    def pack(self):
        """-> (bases uint8[nbytes], UnitTable)"""
        bases = np.zeros(max(self._nbytes, 16), dtype=np.uint8)
        for (m, stride), off in zip(self._mats, self._mat_off):
            n, L = m.shape
            if n and L:
                bases[off:off + n * stride].reshape(n, stride)[:, :L] = m
        descs = np.zeros(len(self._units), dtype=DESC_DTYPE)
        ridx = []
        nri = 0
        for i, (mid, rows, cds) in enumerate(self._units):
            m, stride = self._mats[mid]
            d = descs[i]
            d["base_off"] = self._mat_off[mid]
            d["n_sites"] = m.shape[1]
            d["row_stride"] = stride
            d["mat_rows"] = m.shape[0]
            d["flags"] = PFA_FLAG_CDS if cds else 0
            if rows is None:
                d["n_rows"] = m.shape[0]
                d["row_idx_off"] = PFA_NO_ROW_IDX
            else:
                d["n_rows"] = len(rows)
                d["row_idx_off"] = nri
                ridx.append(rows)
                nri += len(rows)
        row_idx = np.concatenate(ridx) if ridx else np.zeros(0, dtype=np.uint32)
        return bases, finalize_layout(descs, row_idx, len(bases))


def codon_sets(plan_handle, unit: int, n_sites: int, d_bases=None) -> np.ndarray:
    """The set of distinct clean codons of every codon column of one unit of a plan (pfa_plan_codon_sets: a re-scan of
    that unit on the card; `cods` of getvarCDSsites after its regex filter, PolyFastA.py:291, :301): uint64[ceil(L / 3)].
    `d_bases` None: the matrices of the plan's last run."""
    out = np.zeros((int(n_sites) + 2) // 3, dtype=np.uint64)
    p = None if d_bases is None else (d_bases.ptr if isinstance(d_bases, DeviceBuffer) else ctypes.c_void_p(int(d_bases)))
    _lib.check(_lib.load().pfa_plan_codon_sets(plan_handle, p, int(unit), _lib.ptr(out) if len(out) else None, len(out)))
    return out


def resolve_syn_ties(plan_handle, syn_num, n_sites, d_bases=None, every=None, missing_cols=None):
    """Per unit NaN, or — for a unit whose synonymous-site count sits on a two-decimal rounding tie — that count as the
    reference itself accumulates it, column by column in floating point (PolyFastA.py:307), replayed from the unit's codon
    sets (stats.syn_sites_replay / syn_sites_most_likely).  A handful of units in a thousand; everything else prints the same digits either way.
    Call it while the matrices of the plan's run are still on the card.  `every`: bool per unit — replay these whatever
    their bins say (the API mirror returns the reference's count_syn itself).  `missing_cols`: the records' count of codon columns
    without a clean codon — with it, units whose non-synonymous site count is zero in exact arithmetic are replayed as well
    (stats.syn_tie_candidates)."""
    from . import stats
    n_sites = np.asarray(n_sites)
    out = np.full(len(n_sites), np.nan, dtype=np.float64)
    if syn_num is None or len(out) == 0:
        return out
    tie = stats.syn_tie_candidates(syn_num, n_sites, missing_cols)
    pick = tie if every is None else tie | np.asarray(every, dtype=bool)
    for i in np.flatnonzero(pick).tolist():
        sets = codon_sets(plan_handle, i, int(n_sites[i]), d_bases)
        # on a tie the reference's own digit can hang on its interpreter's hash seed (the summation order of a column of three
        # or more codons): then the digits it prints most often (stats.syn_sites_most_likely); the same value otherwise
        out[i] = stats.syn_sites_most_likely(sets, int(n_sites[i])) if tie[i] else stats.syn_sites_replay(sets)
    return out


@dataclass
class ScanOutput:
    table: UnitTable
    results: np.ndarray            # RESULT_DTYPE[n_units]
    sfs_flat: np.ndarray           # uint32
    syn_num: np.ndarray = None     # uint32[n_units, 65] (coding plans)
    sites_s: np.ndarray = None     # float64[n_units]: NaN, or the replayed synonymous-site count of a unit on a rounding tie (scan_batch)
    var_bitmap: np.ndarray = None
    syn_bitmap: np.ndarray = None
    non_bitmap: np.ndarray = None

    def sfs(self, i: int) -> np.ndarray:
        d = self.table.descs[i]
        return self.sfs_flat[int(d["sfs_off"]):int(d["sfs_off"]) + int(d["n_rows"]) // 2]

    def _positions(self, words, i, words_per_tile, tile_sites):
        d = self.table.descs[i]
        nt = int(self.table.n_tiles[i])
        if nt == 0:
            return np.zeros(0, dtype=np.uint32)
        w = np.ascontiguousarray(words[int(d["tile_off"]) * words_per_tile:(int(d["tile_off"]) + nt) * words_per_tile])
        out = np.empty(int(d["n_sites"]), dtype=np.uint32)
        n = _lib.check(_lib.load().pfa_bitmap_positions(_lib.ptr(w), nt, words_per_tile, tile_sites,
                                                        int(d["n_sites"]), _lib.ptr(out), len(out)))
        return out[:n]

    def var_positions(self, i: int) -> np.ndarray:
        """ascending positions of the variable sites of unit i (``pos`` of getvarsites)"""
        return self._positions(self.var_bitmap, i, self.table.var_words, self.table.tile_sites)

    def syn_positions(self, i: int) -> np.ndarray:
        return self._positions(self.syn_bitmap, i, self.table.var_words, self.table.tile_sites)

    def non_positions(self, i: int) -> np.ndarray:
        return self._positions(self.non_bitmap, i, self.table.var_words, self.table.tile_sites)


class Plan:
    """A validated, device-resident descriptor table plus its output buffers."""

    def __init__(self, table: UnitTable, want_pos: bool = False, want_sn: bool = False, timing: bool = False):
        """`timing`: record HIP events around the scan launches of every run (kernel_ms)."""
        _lib.init()
        self.table = table
        self.want = (_lib.PFA_WANT_VAR_BITMAP if want_pos else 0) | (_lib.PFA_WANT_SN_BITMAP if want_sn else 0)
        self.any_cds = bool(((table.descs["flags"] & PFA_FLAG_CDS) != 0).any())
        h = ctypes.c_void_p()
        _lib.check(_lib.load().pfa_plan_create(_lib.ptr(table.descs), table.n_units,
                                               _lib.ptr(table.row_idx) if len(table.row_idx) else None,
                                               len(table.row_idx), table.n_bases_bytes, self.want, ctypes.byref(h)))
        self.handle = h
        self._cache = None
        if timing:
            _lib.check(_lib.load().pfa_plan_set_timing(h, 1))

    def run(self, d_bases, variant: int = 0):
        p = d_bases.ptr if isinstance(d_bases, DeviceBuffer) else ctypes.c_void_p(int(d_bases))
        _lib.check(_lib.load().pfa_plan_run(self.handle, p, int(variant)))

    def kernel_ms(self, n: int) -> np.ndarray:
        buf = (ctypes.c_float * max(n, 1))()
        cnt = _lib.check(_lib.load().pfa_plan_kernel_ms(self.handle, buf, n))
        return np.array(buf[:cnt], dtype=np.float64)

    def _sizes(self):
        if self._cache is None:
            lib = _lib.load()
            self._cache = (int(lib.pfa_plan_n_tiles(self.handle)), max(int(lib.pfa_plan_sfs_words(self.handle)), 1))
        return self._cache

    def _outputs(self):
        t = self.table
        nt, sfs_words = self._sizes()
        res = np.empty(t.n_units, dtype=RESULT_DTYPE)
        sfs = np.empty(sfs_words, dtype=np.uint32)
        if sfs_words == 1:
            sfs[0] = 0
        syn = np.empty((t.n_units, PFA_SYN_BINS), dtype=np.uint32) if self.any_cds else None
        vb = np.zeros(max(nt * t.var_words, 1), dtype=np.uint64) if self.want & _lib.PFA_WANT_VAR_BITMAP else None
        sn = bool(self.want & _lib.PFA_WANT_SN_BITMAP) and self.any_cds
        sb = np.zeros(max(nt * t.var_words, 1), dtype=np.uint64) if sn else None
        nb = np.zeros(max(nt * t.var_words, 1), dtype=np.uint64) if sn else None
        return res, sfs, syn, vb, sb, nb

    def fetch(self) -> ScanOutput:
        res, sfs, syn, vb, sb, nb = self._outputs()
        _lib.check(_lib.load().pfa_plan_fetch(self.handle, _lib.ptr(res), _lib.ptr(sfs), _lib.ptr(syn), _lib.ptr(vb),
                                              _lib.ptr(sb), _lib.ptr(nb)))
        return ScanOutput(table=self.table, results=res, sfs_flat=sfs, syn_num=syn, var_bitmap=vb, syn_bitmap=sb, non_bitmap=nb)

    def run_fetch(self, d_bases, variant: int = 0) -> ScanOutput:
        """run + fetch in one call into the library (one scan per call: the latency path)"""
        p = d_bases.ptr if isinstance(d_bases, DeviceBuffer) else ctypes.c_void_p(int(d_bases))
        res, sfs, syn, vb, sb, nb = self._outputs()
        _lib.check(_lib.load().pfa_plan_run_fetch(self.handle, p, int(variant), _lib.ptr(res), _lib.ptr(sfs), _lib.ptr(syn),
                                                  _lib.ptr(vb), _lib.ptr(sb), _lib.ptr(nb)))
        return ScanOutput(table=self.table, results=res, sfs_flat=sfs, syn_num=syn, var_bitmap=vb, syn_bitmap=sb, non_bitmap=nb)

    def codon_sets(self, unit: int, d_bases=None) -> np.ndarray:
        return codon_sets(self.handle, unit, int(self.table.descs[unit]["n_sites"]), d_bases)

    def results_dev(self):
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _lib.check(_lib.load().pfa_plan_results_dev(self.handle, ctypes.byref(p), ctypes.byref(n)))
        return p.value, n.value

    @property
    def description(self) -> str:
        """one entry per kernel launch of a run (bucket), largest first"""
        return (_lib.load().pfa_plan_describe(self.handle) or b"").decode()

    @property
    def n_launches(self) -> int:
        return int(_lib.load().pfa_plan_n_launches(self.handle))

    @property
    def tile_width(self) -> int:
        return int(_lib.load().pfa_plan_tile_width(self.handle))

    @property
    def tiles_per_chunk(self) -> int:
        return int(_lib.load().pfa_plan_tiles_per_chunk(self.handle))

    @property
    def lds_bytes(self) -> int:
        return int(_lib.load().pfa_plan_lds_bytes(self.handle))

    @property
    def rows_per_chunk(self) -> int:
        """rows staged per pass by the row-chunked kernel (0: the whole tile at once)"""
        return int(_lib.load().pfa_plan_rows_per_chunk(self.handle))

    def close(self):
        if self.handle is not None and self.handle.value:
            _lib.load().pfa_plan_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def scan_batch(batch: Batch, want_pos: bool = False, want_sn: bool = False, variant: int = 0, exact_syn: bool = False) -> ScanOutput:
    """Pack, upload, scan, fetch: the one-call path used by the API mirror and the CLI.  `exact_syn`: `sites_s` of EVERY
    coding unit is the reference's own floating-point accumulation (else only of units on a rounding tie)."""
    bases, table = batch.pack()
    if table.n_units == 0:
        raise ValueError("empty batch")
    dev = DeviceBuffer(len(bases))
    try:
        dev.upload(bases)
        plan = Plan(table, want_pos=want_pos, want_sn=want_sn)
        try:
            plan.run(dev, variant=variant)
            out = plan.fetch()
            if out.syn_num is not None:
                coding = (table.descs["flags"] & PFA_FLAG_CDS) != 0
                out.sites_s = resolve_syn_ties(plan.handle, np.where(coding[:, None], out.syn_num, 0), table.descs["n_sites"],
                                               every=coding if exact_syn else None, missing_cols=out.results["missing_cols"])
            return out
        finally:
            plan.close()
    finally:
        dev.free()


def scan_matrix(matrix: np.ndarray, rows=None, cds: bool = False, want_pos: bool = False,
                want_sn: bool = False, variant: int = 0, exact_syn: bool = False) -> ScanOutput:
    b = Batch()
    b.add_unit(b.add_matrix(matrix), rows=rows, cds=cds)
    return scan_batch(b, want_pos=want_pos, want_sn=want_sn, variant=variant, exact_syn=exact_syn)
