#!/usr/bin/env python3
"""Timeline of one files -> CSV run (PFA_TRACE): tools/e2e_trace.py [files] [extra CLI flags...]"""
import io, os, shutil, sys, tempfile, time
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, dist, synth  # noqa: E402
# The files are written by this thread: bind it to the CPUs next to the GPU FIRST (as bench.py does), or the tmpfs pages land on whichever
# socket the scheduler picked and every run of this process reads them across the socket link — 143 k instead of 220 k alignments/s, in about
# half of the processes (round 4: the "slow mode" of a whole process).  PFA_NO_BIND=1: leave it to chance.
if not os.environ.get("PFA_NO_BIND"):
    print("bound to", len(dist.bind_to_gpu_node(0, 1) or []), "CPUs", flush=True)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
extra = sys.argv[2:]
N, L = 50, 10000
_lib.init(0)
p = synth.make_params(N, 0.01)
d = tempfile.mkdtemp(prefix="pfa_trace_", dir="/dev/shm")
try:
    hdr = [f">{h}\n".encode() for h in synth.headers_for(N, 2)]
    base = [synth.synth_alignment(11, k, L, p) for k in range(32)]
    for m in range(M):
        mat = base[m % 32]
        with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
            fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
    out = os.path.join(tempfile.gettempdir(), f"pfa_trace_{os.getpid()}.csv")
    times = []
    R = int(os.environ.get("PFA_REPS", "7"))
    for rep in range(R):
        if os.path.exists(out):
            os.unlink(out)
        os.environ.pop("PFA_TRACE", None)
        if rep == R - 1 and os.environ.get("PFA_TRACE_LAST"):
            os.environ["PFA_TRACE"] = "1"
        t0 = time.perf_counter()
        with redirect_stdout(io.StringIO()):
            cli.main(["-d", d, "-o", out, "-s"] + extra)
        times.append(time.perf_counter() - t0)
    warm = sorted(times[1:])
    gb = M * N * (L + 10) / 1e9
    print(f"first {times[0] * 1e3:7.1f} ms | warm best {warm[0] * 1e3:6.1f} ms = {M / warm[0]:7.0f} aln/s = {gb / warm[0]:5.1f} GB/s | "
          f"median {warm[len(warm) // 2] * 1e3:6.1f} ms = {M / warm[len(warm) // 2]:7.0f} aln/s | worst {warm[-1] * 1e3:6.1f} ms", flush=True)
    os.unlink(out)
finally:
    shutil.rmtree(d, ignore_errors=True)
