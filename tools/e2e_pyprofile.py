#!/usr/bin/env python3
"""Where the Python side of one warm files -> CSV run spends its time (cProfile): tools/e2e_pyprofile.py [files] [extra CLI flags...]"""
import cProfile, io, os, pstats, shutil, sys, tempfile, time
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, dist, synth  # noqa: E402
dist.bind_to_gpu_node(0, 1)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
extra = sys.argv[2:]
N, L = 50, 10000
_lib.init(0)
p = synth.make_params(N, 0.01)
d = tempfile.mkdtemp(prefix="pfa_prof_", dir="/dev/shm")
try:
    hdr = [f">{h}\n".encode() for h in synth.headers_for(N, 2)]
    base = [synth.synth_alignment(11, k, L, p) for k in range(32)]
    for m in range(M):
        mat = base[m % 32]
        with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
            fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
    out = os.path.join(tempfile.gettempdir(), f"pfa_prof_{os.getpid()}.csv")
    for rep in range(4):
        if os.path.exists(out):
            os.unlink(out)
        pr = cProfile.Profile() if rep == 3 else None
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            if pr:
                pr.enable()
            cli.main(["-d", d, "-o", out, "-s"] + extra)
            if pr:
                pr.disable()
        print(f"run {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
    print(s.getvalue())
finally:
    shutil.rmtree(d, ignore_errors=True)
