#!/usr/bin/env python3
"""Where and when the workgroups of scan_kernel's last launch ran (PFA_WAVE_TRACE=<file> makes the library write the file):
workgroups per XCD and per CU, their start times, pieces and tiles scanned.  usage: tools/wave_trace.py FILE"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint32).reshape(-1, 8)
a = a[a[:, 7] == 1]
hw, xcc = a[:, 0], a[:, 1] & 0xF
t0 = (a[:, 2].astype(np.uint64) | (a[:, 3].astype(np.uint64) << 32)).astype(np.float64) / 100.0      # us (100 MHz)
dur = a[:, 4].astype(np.float64) / 100.0
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
simd = (hw >> 4) & 0x3
print("workgroups", len(a), "pieces", int(a[:, 5].sum()), "tiles", int(a[:, 6].sum()))
t0 -= t0.min()
print("start us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(t0, [0, 50, 90, 99, 100])))
end = t0 + dur
print("end   us: p0 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(end, [0, 50, 90, 99, 100])))
print("tiles per workgroup: min %d p50 %d max %d" % (a[:, 6].min(), np.median(a[:, 6]), a[:, 6].max()))
for x in range(8):
    m = xcc == x
    if not m.any():
        continue
    key = (se[m].astype(np.int64) << 8) | (sh[m].astype(np.int64) << 4) | cu[m]
    u, c = np.unique(key, return_counts=True)
    print(f"xcc {x}: workgroups {m.sum()} CUs {len(u)} per-CU min {c.min()} max {c.max()} hist {np.bincount(c).tolist()}  tiles {int(a[m, 6].sum())}  last end {end[m].max():.1f} us  late starters (>5us) {(t0[m] > 5).sum()}")
    su, sc = np.unique((key << 2) | simd[m], return_counts=True)
    print(f"        per-SIMD waves hist {np.bincount(sc).tolist()}")
