#!/usr/bin/env python3
"""Latency of small launches by tile width (PFA_FORCE_TW): one 50 x 10 kb alignment per launch,
and BASELINE config 3 (200 x 100 kb, two populations) as one launch.  Kernel time from HIP events
and wall time of run + fetch.  usage: tools/small_launch_probe.py > gpurun_out/small_probe.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from polyfasta_amd import _lib, scan  # noqa: E402


def probe(w, label):
    lib = _lib.load()
    for tw in ("", "256", "128", "64"):
        if tw:
            os.environ["PFA_FORCE_TW"] = tw
        else:
            os.environ.pop("PFA_FORCE_TW", None)
        plan = scan.Plan(w.table)
        for _ in range(30):
            plan.run_fetch(w.dev)
        t0 = time.perf_counter()
        for _ in range(300):
            plan.run_fetch(w.dev)
        wall = (time.perf_counter() - t0) / 300
        _lib.check(lib.pfa_plan_set_timing(plan.handle, 1))
        for _ in range(60):
            plan.run_fetch(w.dev)
        k = plan.kernel_ms(50)
        print(f"{label} tw={tw or 'auto':4s} kernel_us avg {k.mean() * 1e3:7.2f} min {k.min() * 1e3:7.2f}  run+fetch {wall * 1e6:7.2f} us  "
              f"{w.alg_bytes / (k.mean() * 1e-3) / 1e9:8.1f} GB/s  {plan.description}", flush=True)
        plan.close()


def main():
    _lib.init(0)
    args = bench.parse([])
    w1 = bench.Workload(50, 10000, 1, timing=False)
    probe(w1, "single 50x10k")
    w1.close()
    w3 = bench.Workload(100, 100000, 1, pops=2, timing=False)
    probe(w3, "config3 2x100x100k")
    w3.close()
    w4 = bench.Workload(50, 10000, 64, timing=False)
    probe(w4, "64 x 50x10k")
    w4.close()


if __name__ == "__main__":
    main()
