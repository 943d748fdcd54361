#!/usr/bin/env python3
"""Files -> CSV against the CPU quota of the box: loader threads x wait policy, one fresh process per setting (the wait
policy is fixed when the device is initialised), the same tmpfs directory for all, alternating order, the control group's
throttle counters around the warm runs of every setting.

    python tools/e2e_cpu_budget.py [--files 16384] [--settings "16:spin 14:spin 14:block 12:spin"] [--rounds 2] [--flags "..."]
"""
import argparse
import ctypes
import io
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def throttle():
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0)), int(d.get("usage_usec", 0))
    except OSError:
        return 0, 0, 0


def child(directory, flags, reps):
    from contextlib import redirect_stdout
    from polyfasta_amd import cli
    out = os.path.join(tempfile.gettempdir(), f"pfa_budget_{os.getpid()}.csv")
    times = []
    for rep in range(1 + reps):
        if os.path.exists(out):
            os.unlink(out)
        if rep == 1:
            th0 = throttle()
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            rc = cli.main(["-d", directory, "-o", out, "-s"] + flags)
        times.append(time.perf_counter() - t0)
        assert rc == 0
    th1 = throttle()
    n = sum(1 for _ in open(out))
    os.unlink(out)
    warm = sorted(times[1:])
    st = cli.last_stats
    print(json.dumps({"threads": st["loader_threads"], "best_s": warm[0], "median_s": warm[len(warm) // 2], "first_s": times[0], "rows": n,
                      "throttled_periods": th1[0] - th0[0], "throttled_ms": (th1[1] - th0[1]) / 1e3, "cpu_ms": (th1[2] - th0[2]) / 1e3,
                      "wall_ms_warm": sum(times[1:]) * 1e3}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=16384)
    ap.add_argument("--settings", default="16:spin 14:spin 14:block 12:spin auto:auto")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--flags", default="")
    ap.add_argument("--sites", type=int, default=10000)
    ap.add_argument("--child", default=None)
    ap.add_argument("--write-from", default="gpu", choices=["gpu", "other", "unbound"],
                    help="which socket's CPUs write the files (= where their tmpfs pages land): the GPU's, the other one, wherever the scheduler puts this process")
    a = ap.parse_args()
    flags = a.flags.split() if a.flags else []
    if a.child:
        return child(a.child, flags, a.reps)
    import numpy as np
    from polyfasta_amd import _lib, dist, scan, synth
    # the files' tmpfs pages belong next to the GPU, where the children (which bind themselves there) read them: written from
    # the other socket they are read across the socket link, 145-150 k alignments/s instead of 200-215 k (PFA_WRITE_UNBOUND=1: that case)
    if a.write_from == "gpu":
        dist.bind_to_gpu_node(0, 1)
        os.environ.pop("PFA_CPUS_ARE_MINE", None)
    elif a.write_from == "other":                          # the pages land on the socket the GPU does NOT hang off
        near = set(dist.gpu_local_cpus(0) or [])
        far = sorted(os.sched_getaffinity(0) - near)
        if near and far:
            os.sched_setaffinity(0, far)
    lib = _lib.load()
    N, L, M = 50, a.sites, a.files
    coding = "--cds" in flags
    root = tempfile.mkdtemp(prefix="pfa_budget_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        p = synth.make_params(N, 0.01)
        stride = scan.row_stride_for(L, 128)
        hdr = [f">{h}\n".encode() for h in synth.headers_for(N, 2 if "-p" in flags else 0)]
        for m0 in range(0, M, 2048):
            mc = min(2048, M - m0)
            sp = _lib.SynthParamsC(seed=7, first_aln=m0, n_aln=mc, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                                   tri_thresh=p.tri_thresh, n_units=len(p.units), coding=int(coding))
            bases = np.zeros(mc * N * stride, dtype=np.uint8)
            _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)), _lib.ptr(np.ascontiguousarray(p.units))))
            for m in range(mc):
                mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
                with open(os.path.join(root, f"aln{m0 + m:06d}.fa"), "wb") as fh:
                    fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
        print(f"# {M} files of {N} x {L} in {root}; flags: {' '.join(flags) or '(plain)'}; cpu.max: "
              f"{open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else 'n/a'}; "
              f"CPUs: {len(os.sched_getaffinity(0))}; files written from: {a.write_from}", flush=True)
        for rnd in range(a.rounds):
            for s in a.settings.split():
                thr, wait, pin = (s.split(":") + ["0"])[:3]     # threads : wait policy [: core stride of pinned loaders, PFA_PIN_LOADERS]
                env = dict(os.environ)
                env.pop("PFA_THREADS", None)
                env.pop("PFA_WAIT", None)
                env.pop("PFA_PIN_LOADERS", None)
                if "=" in pin:                                   # VAR=x+VAR2=y instead of a pin setting
                    env.update(kv.split("=", 1) for kv in pin.split("+"))
                elif pin != "0":
                    env["PFA_PIN_LOADERS"] = pin
                if thr != "auto":
                    env["PFA_THREADS"] = thr
                if wait != "auto":
                    env["PFA_WAIT"] = wait
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", root, "--reps", str(a.reps), "--flags", a.flags],
                                    capture_output=True, text=True, env=env, timeout=900)
                if pr.returncode != 0:
                    print(f"round {rnd} {s}: FAILED rc={pr.returncode} {pr.stderr[-400:]}", flush=True)
                    continue
                j = json.loads(pr.stdout.strip().splitlines()[-1])
                print(f"round {rnd} {s:>34s}: loaders {j['threads']:2d}  best {M / j['best_s'] / 1e3:6.1f} k/s  median {M / j['median_s'] / 1e3:6.1f} k/s  first {j['first_s']:.3f} s  "
                      f"throttled {j['throttled_periods']:3d} periods / {j['throttled_ms']:8.1f} ms  cpu {j['cpu_ms']:7.0f} ms over {j['wall_ms_warm']:6.0f} ms of wall = "
                      f"{j['cpu_ms'] / max(j['wall_ms_warm'], 1e-9):5.1f} CPUs", flush=True)
    finally:
        import shutil
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
