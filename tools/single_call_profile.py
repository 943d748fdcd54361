#!/usr/bin/env python3
"""Where does one single-alignment CLI call spend its time? (config 2 shape, in-process, warm)"""
import cProfile, io, os, pstats, sys, tempfile, time
from contextlib import redirect_stdout
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, synth
_lib.init(0)
p = synth.make_params(50, 0.01)
m = synth.synth_alignment(802, 0, 10000, p)
d = tempfile.mkdtemp(prefix="pfa_one_", dir="/dev/shm")
path = os.path.join(d, "x.fa")
open(path, "w").write(synth.to_fasta_text(synth.headers_for(50), m))
def run():
    with redirect_stdout(io.StringIO()):
        cli.main(["-f", path, "-s"])
for _ in range(5): run()
t0 = time.perf_counter()
for _ in range(50): run()
print("per call: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): run()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
