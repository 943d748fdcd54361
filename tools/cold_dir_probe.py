#!/usr/bin/env python3
"""A whole directory through a FRESH process (`python -m polyfasta_amd -d DIR -o out.csv -s`): wall time from spawn to exit, the
start-up marks and the per-batch timeline (PFA_TRACE) of the run — what a user's one job pays beyond the warm rate of bench.py.
usage: tools/cold_dir_probe.py [files] [runs]"""
import ctypes
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from polyfasta_amd import _lib, scan, synth  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, L = 50, 10000
lib = _lib.load()
root = tempfile.mkdtemp(prefix="pfa_cold_dir_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    p = synth.make_params(N, 0.01)
    stride = scan.row_stride_for(L, 128)
    hdr = [f">{h}\n".encode() for h in synth.headers_for(N)]
    for m0 in range(0, M, 2048):
        mc = min(2048, M - m0)
        sp = _lib.SynthParamsC(seed=7, first_aln=m0, n_aln=mc, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                               tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
        bases = np.zeros(mc * N * stride, dtype=np.uint8)
        _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)), _lib.ptr(np.ascontiguousarray(p.units))))
        for m in range(mc):
            mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
            with open(os.path.join(root, f"aln{m0 + m:06d}.fa"), "wb") as fh:
                fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
    out = os.path.join(tempfile.gettempdir(), f"pfa_cold_dir_{os.getpid()}.csv")
    for r in range(runs):
        if os.path.exists(out):
            os.unlink(out)
        env = dict(os.environ, PFA_TRACE="1", PFA_TRACE_STARTUP="1")
        for k in sys.argv[3:]:
            a, b = k.split("=", 1)
            env[a] = b
        def throttle():
            try:
                d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
                return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)) / 1e3, int(d.get("usage_usec", 0)) / 1e3
            except OSError:
                return 0, 0.0, 0.0
        th0 = throttle()
        t0w, t0 = time.time(), time.perf_counter()
        pr = subprocess.run([sys.executable, "-m", "polyfasta_amd", "-d", root, "-o", out, "-s"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        wall = time.perf_counter() - t0
        rows = sum(1 for _ in open(out)) if os.path.exists(out) else 0
        th1 = throttle()
        print(f"run {r}: rc {pr.returncode}  wall {wall * 1e3:7.1f} ms  rows {rows}  ({M / wall:8.0f} files/s whole process)  control group: throttled "
              f"{th1[0] - th0[0]} periods / {th1[1] - th0[1]:.0f} ms, CPU {th1[2] - th0[2]:.0f} ms")
        import json
        for ln in pr.stderr.splitlines():
            if ln.startswith("PFA_STARTUP "):
                prev = t0w
                for ph, t in json.loads(ln[len("PFA_STARTUP "):]):
                    print(f"      {ph:62s} {1e3 * (t - prev):7.1f} ms")
                    prev = t
                print(f"      {'exit':62s} {1e3 * (t0w + wall - prev):7.1f} ms")
        for ln in pr.stderr.splitlines():
            if ln.startswith("PFA_EXIT"):
                print("      " + ln)
        tr = [ln for ln in pr.stderr.splitlines() if ln.startswith("PFA_TRACE")]
        for ln in tr[:2] + [x for x in tr if " batch " in x][:12] + [x for x in tr if "loop_end" in x or "close" in x or "run_files" in x]:
            print("      " + ln[:260])
finally:
    shutil.rmtree(root, ignore_errors=True)
