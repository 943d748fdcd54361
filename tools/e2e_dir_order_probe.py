#!/usr/bin/env python3
"""Why does bench.py's `plain` end-to-end leg (50 x 10 000 files, written and read first) read 10 % fewer bytes per second than the
legs on the 50 x 9 000 directory?  Directories of several row lengths, written in a given order, each read by the plain command line
(in-process, 1 + 4 runs): files/s, text GB/s and read() GB/s per loader thread.  usage: tools/e2e_dir_order_probe.py M L1,L2,...  [L1,L2,... again in another order]"""
import ctypes
import io
import os
import shutil
import sys
import tempfile
import time
from contextlib import redirect_stdout

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, dist, scan, synth  # noqa: E402

M = int(sys.argv[1])
orders = [[int(x) for x in a.split(",")] for a in sys.argv[2:]]
dist.bind_to_gpu_node(0, 1)
_lib.init()
lib = _lib.load()
N = 50
for order in orders:
    root = tempfile.mkdtemp(prefix="pfa_order_", dir="/dev/shm")
    out = os.path.join(tempfile.gettempdir(), f"pfa_order_{os.getpid()}.csv")
    try:
        dirs = []
        for L in order:
            d = os.path.join(root, f"L{L}_{len(dirs)}")
            os.makedirs(d)
            p = synth.make_params(N, 0.01)
            stride = scan.row_stride_for(L, 128)
            hdr = [f">indiv{i + 1}\n".encode() for i in range(N)]
            for m0 in range(0, M, 2048):
                mc = min(2048, M - m0)
                sp = _lib.SynthParamsC(seed=7, first_aln=m0, n_aln=mc, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                                       tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
                bases = np.zeros(mc * N * stride, dtype=np.uint8)
                _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)),
                                                   _lib.ptr(np.ascontiguousarray(p.units))))
                for m in range(mc):
                    mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
                    with open(os.path.join(d, f"aln{m0 + m:06d}.fa"), "wb") as fh:
                        fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
            dirs.append((L, d))
        print(f"written in the order {order}, {M} files each", flush=True)
        for rnd in range(2):
            for L, d in dirs:
                times, best_stats = [], None
                for rep in range(5):
                    if os.path.exists(out):
                        os.unlink(out)
                    t0 = time.perf_counter()
                    with redirect_stdout(io.StringIO()):
                        cli.main(["-d", d, "-o", out, "-s"])
                    times.append(time.perf_counter() - t0)
                    if rep and times[-1] == min(times[1:]):
                        best_stats = dict(cli.last_stats)
                best = min(times[1:])
                fb = os.path.getsize(os.path.join(d, "aln000000.fa"))
                nt = max(best_stats["loader_threads"], 1)
                print(f"  round {rnd} L {L:6d} file {fb:7d} B: best {M / best / 1e3:7.1f} k files/s  {M * fb / best / 1e9:6.1f} GB/s of text   read() "
                      f"{best_stats['bytes_read'] / (best_stats['read_thread_s'] / nt) / 1e9:6.1f} GB/s all threads  pack {best_stats['bytes_read'] / (best_stats['index_thread_s'] / nt) / 1e9:6.1f}"
                      f"  median {M / sorted(times[1:])[2] / 1e3:7.1f} k", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
        if os.path.exists(out):
            os.unlink(out)
