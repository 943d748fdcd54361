#!/usr/bin/env python3
"""CPU time (control group's usage_usec) and wall of short commands: who burns the CPU quota of a fresh one-job process?"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stat():
    d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
    return int(d["usage_usec"]) / 1e3, int(d.get("nr_throttled", 0))


def run(label, cmd, env=None, n=3):
    for _ in range(n):
        c0, t0 = stat(), time.perf_counter()
        subprocess.run(cmd, env=dict(os.environ, **(env or {})), capture_output=True, cwd=ROOT)
        wall = (time.perf_counter() - t0) * 1e3
        c1 = stat()
        print(f"{label:58s} wall {wall:7.1f} ms  CPU {c1[0] - c0[0]:7.1f} ms  throttled periods {c1[1] - c0[1]}", flush=True)


py = sys.executable
run("python -c pass", [py, "-c", "pass"])
run("python -c 'import numpy'", [py, "-c", "import numpy"])
run("python -c 'import numpy' OPENBLAS_NUM_THREADS=1", [py, "-c", "import numpy"], {"OPENBLAS_NUM_THREADS": "1"})
run("python -c 'import numpy' OMP_NUM_THREADS=1", [py, "-c", "import numpy"], {"OMP_NUM_THREADS": "1"})
run("init_probe (hipInit + stream + first launch)", [os.path.join(ROOT, "tools/ubench/init_probe")])
run("python -c 'ctypes: pfa_init only'", [py, "-c", "import sys; sys.path.insert(0, %r); from polyfasta_amd import _lib; _lib.init(0)" % ROOT])
run("python -m polyfasta_amd -f one file", [py, "-m", "polyfasta_amd", "-f", os.path.join(ROOT, "tests/golden/example_theta_0.01/file1.fa")])
run("python -m polyfasta_amd -f one file OPENBLAS_NUM_THREADS=1", [py, "-m", "polyfasta_amd", "-f", os.path.join(ROOT, "tests/golden/example_theta_0.01/file1.fa")], {"OPENBLAS_NUM_THREADS": "1"})
