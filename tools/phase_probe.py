#!/usr/bin/env python3
"""Timing experiment (needs a library built with -DPFA_PHASES=1, see profiles/r05_tall_phase_timers.patch): shader cycles per phase of
scan_tall_kernel's chunk loop.  usage: PFA_LIB=... tools/phase_probe.py --rows 5000 --sites 9000 --batch 1024 [--coding]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402

NAMES = ["wait for the chunk (vmcnt)", "tier 1", "new count slots", "new codon slots", "tier 2", "codon slots' rows", "request the next chunk",
         "end of tile (H, SFS, column path, codon pass)", "loop top", "-"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=5000)
    ap.add_argument("--sites", type=int, default=9000)
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--coding", action="store_true")
    a = ap.parse_args()
    from polyfasta_amd import _lib
    _lib.init(0)
    lib = _lib.load()
    w = bench.Workload(a.rows, a.sites, a.batch, coding=a.coding)
    for _ in range(3):
        w.plan.run(w.dev)
    _lib.check(lib.pfa_sync())
    out = (ctypes.c_ulonglong * 16)()
    lib.pfa_debug_phases.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    assert lib.pfa_debug_phases(out, 1) == 0
    n = 5
    for _ in range(n):
        w.plan.run(w.dev)
    _lib.check(lib.pfa_sync())
    ms = float(np.mean(w.plan.kernel_ms(1)))
    assert lib.pfa_debug_phases(out, 1) == 0
    v = np.array(list(out), dtype=np.float64)
    chunks, waves = v[14], v[15]
    total = v[:10].sum()
    print(json.dumps({"workload": w.label(), "kernel": w.plan.description, "kernel_ms_with_timers": ms, "wavefronts": waves / n, "chunks": chunks / n}))
    print(f"cycles per chunk and wavefront: {total / chunks:8.0f}")
    for i in range(9):
        print(f"  {NAMES[i]:48s} {v[i] / chunks:8.0f}  {v[i] / total:6.3f}")


if __name__ == "__main__":
    main()
