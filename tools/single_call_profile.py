#!/usr/bin/env python3
"""One alignment per call, 300 times (under rocprofv3 --kernel-trace --stats: the kernel durations of
the latency path, without the HIP-event overhead).  usage: ... -- python3 tools/single_call_profile.py [config3]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from polyfasta_amd import _lib, scan  # noqa: E402

_lib.init(0)
if len(sys.argv) > 1 and sys.argv[1] == "config3":
    w = bench.Workload(100, 100000, 1, pops=2, timing=False)
else:
    w = bench.Workload(50, 10000, 1, timing=False)
for _ in range(300):
    w.plan.run_fetch(w.dev)
print(w.plan.description)
w.close()
