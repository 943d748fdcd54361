#!/usr/bin/env python3
"""cProfile of `python -m polyfasta_amd -f <M files> -o out.csv -s` (where does the host time go?)."""
import cProfile, io, os, pstats, sys, tempfile, time, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, scan, synth
M, N, L = int(sys.argv[1]) if len(sys.argv) > 1 else 4000, 50, 10000
lib = _lib.load(); _lib.init(0)
p = synth.make_params(N, 0.01)
stride = scan.row_stride_for(L, 128)
sp = _lib.SynthParamsC(seed=7, first_aln=0, n_aln=M, n_rows=N, n_sites=L, row_stride=stride,
                       seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
bases = np.zeros(M * N * stride, dtype=np.uint8)
_lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                   _lib.ptr(np.ascontiguousarray(p.units))))
d = tempfile.mkdtemp(prefix="pfa_prof_", dir="/dev/shm")
hdr = [f">indiv{i + 1}\n".encode() for i in range(N)]
for m in range(M):
    mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
    with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
        fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
out = os.path.join(d, "_out.csv")
argv = ["-d", d, "-o", out, "-s"]
cli.main(argv); os.unlink(out)
t0 = time.perf_counter(); cli.main(argv); dt = time.perf_counter() - t0; os.unlink(out)
print(f"{M} files: {dt * 1e3:.1f} ms = {M / dt:.0f} alignments/s")
pr = cProfile.Profile(); pr.enable(); cli.main(argv); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
import shutil; shutil.rmtree(d)
