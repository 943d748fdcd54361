"""ctypes front end of oracle/pfa_oracle.c (TEST INFRASTRUCTURE ONLY).

The genetic-code tables handed to the C code are derived here from the pinned
Python restatement (oracle/polyfasta_oracle.py), so the C oracle shares no
table code with the product.  Codon index = 16*e0 + 4*e1 + e2, A=0 C=1 G=2 T=3.
"""
import ctypes
import os
import subprocess

import numpy as np

from . import polyfasta_oracle as orc

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpfa_oracle.so")
_BASES = "ACGT"


class OracleOut(ctypes.Structure):
    _fields_ = [("H", ctypes.c_uint64), ("H_S", ctypes.c_uint64), ("H_N", ctypes.c_uint64),
                ("S", ctypes.c_uint32), ("sfs_bad", ctypes.c_uint32), ("seg_S", ctypes.c_uint32),
                ("seg_N", ctypes.c_uint32), ("nstops", ctypes.c_uint32), ("missing_cols", ctypes.c_uint32),
                ("count_syn", ctypes.c_double)]


class OracleTables(ctypes.Structure):
    _fields_ = [("pair", ctypes.c_void_p), ("annot_id", ctypes.c_void_p), ("syn3", ctypes.c_void_p)]


def build(force=False):
    src = os.path.join(_HERE, "pfa_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpfa_oracle.so"])
    return _SO


def codon_str(idx):
    return _BASES[idx >> 4] + _BASES[(idx >> 2) & 3] + _BASES[idx & 3]


def _make_tables():
    pair = np.zeros(64 * 64, dtype=np.uint8)
    annot = np.full(64, 255, dtype=np.uint8)
    syn3 = np.zeros(64, dtype=np.uint8)
    ids = {}
    for c in range(64):
        s = codon_str(c)
        syn3[c] = orc.syn_thirds(s)
        if s in orc.ANNOT:
            annot[c] = ids.setdefault(orc.ANNOT[s], len(ids))
    for a in range(64):
        for b in range(64):
            sa, sb = codon_str(a), codon_str(b)
            if a != b and sa in orc.ANNOT and sb in orc.ANNOT:
                syn, non = orc.get_syn_nonsyn_cod_sites([[sa, sb], 0])
                pair[a * 64 + b] = sum(1 << p for p in syn) | sum(8 << p for p in non)
    return pair, annot, syn3


_lib = None
_tables = None


def lib():
    global _lib, _tables
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.pfa_oracle_scan.restype = ctypes.c_int
        _lib.pfa_oracle_scan_many.restype = ctypes.c_uint64
        arrays = _make_tables()
        t = OracleTables(*(a.ctypes.data for a in arrays))
        _tables = (t, arrays)  # keep the arrays alive
    return _lib


def scan(matrix, rows=None, cds=False, want_sites=False):
    """Integer summary of one unit.  `matrix`: uint8 [rows, sites] (any row
    stride); `rows`: optional row subset.  Returns a dict."""
    L = lib()
    m = np.ascontiguousarray(matrix, dtype=np.uint8)
    n_mat_rows, n_sites = m.shape
    if rows is None:
        n_rows, rows_p = n_mat_rows, None
    else:
        rows_arr = np.ascontiguousarray(rows, dtype=np.uint32)
        n_rows, rows_p = len(rows_arr), rows_arr.ctypes.data_as(ctypes.c_void_p)
    out = OracleOut()
    sfs = np.zeros(max(n_rows // 2, 1), dtype=np.uint32)
    hsite = np.zeros(max(n_sites, 1), dtype=np.uint32) if want_sites else None
    labels = np.zeros(max(n_sites, 1), dtype=np.uint8) if want_sites else None
    L.pfa_oracle_scan(m.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n_rows), ctypes.c_uint32(n_sites),
                      ctypes.c_uint64(m.strides[0] if n_mat_rows else max(n_sites, 1)), rows_p, ctypes.c_int(int(cds)),
                      ctypes.byref(_tables[0]), ctypes.byref(out), sfs.ctypes.data_as(ctypes.c_void_p),
                      hsite.ctypes.data_as(ctypes.c_void_p) if want_sites else None,
                      labels.ctypes.data_as(ctypes.c_void_p) if want_sites else None)
    res = {k: getattr(out, k) for k, _ in OracleOut._fields_}
    res["sfs"] = sfs[: n_rows // 2].copy()
    if want_sites:
        res["h_site"] = hsite[:n_sites]
        res["labels"] = labels[:n_sites]
    return res


def scan_many(bases, n_mat, n_rows, n_sites, row_stride):
    """Non-coding scan of consecutive matrices (bench cpu_baseline); returns (S[], H[])."""
    L = lib()
    s = np.zeros(n_mat, dtype=np.uint32)
    h = np.zeros(n_mat, dtype=np.uint64)
    L.pfa_oracle_scan_many(bases.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint32(n_mat), ctypes.c_uint32(n_rows),
                           ctypes.c_uint32(n_sites), ctypes.c_uint64(row_stride),
                           s.ctypes.data_as(ctypes.c_void_p), h.ctypes.data_as(ctypes.c_void_p))
    return s, h
