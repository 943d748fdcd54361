"""Timing probe: how long does the result fetch (one pinned D2H + copies) take after a
burst of scans?  (bench.py's `fetch` figure grew out of proportion above 8 192 units.)"""
import sys, time, ctypes, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from polyfasta_amd import _lib, scan, synth
lib = _lib.load(); _lib.init(0)
N, L = 50, 10000
stride = scan.row_stride_for(L, 128)
for B in (8192, 16384, 8192, 16384):
    nbytes = B * N * stride
    dev = scan.DeviceBuffer(nbytes)
    descs = np.zeros(B, dtype=_lib.DESC_DTYPE)
    descs["base_off"] = np.arange(B, dtype=np.uint64) * np.uint64(N * stride)
    descs["n_rows"], descs["n_sites"], descs["row_stride"], descs["mat_rows"] = N, L, stride, N
    descs["row_idx_off"] = _lib.PFA_NO_ROW_IDX
    plan = scan.Plan(scan.finalize_layout(descs, None, nbytes))
    for rep in range(3):
        plan.run(dev); plan.fetch()
    for burst in (1, 5, 20, 50):
        worst = 0.0
        for rep in range(6):
            for _ in range(burst):
                plan.run(dev)
            lib.pfa_sync()
            t0 = time.perf_counter(); out = plan.fetch(); t1 = time.perf_counter()
            worst = max(worst, t1 - t0)
        print(B, "burst", burst, "fetch() last %.3f ms  worst of 6 %.3f ms" % ((t1 - t0) * 1e3, worst * 1e3), flush=True)
    plan.close(); dev.free()
