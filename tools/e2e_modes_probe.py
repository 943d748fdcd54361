#!/usr/bin/env python3
"""End-to-end rate (FASTA files in tmpfs -> CSV) of the CLI modes: plain, --cds --jc, --pops, --pops --cds --jc, --sfs.
usage: tools/e2e_modes_probe.py [files]"""
import io
import os
import shutil
import sys
import tempfile
import time
from contextlib import redirect_stdout

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from polyfasta_amd import _lib, cli, synth  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N, L = 50, 9000
_lib.init(0)
p = synth.make_params(N, 0.01)
d = tempfile.mkdtemp(prefix="pfa_e2e_modes_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    hdr = [f">{h}\n".encode() for h in synth.headers_for(N, 2)]
    base = [synth.synth_alignment(11, k, L, p, coding=True) for k in range(32)]
    for m in range(M):
        mat = base[m % 32]
        with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
            fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
    out = os.path.join(tempfile.gettempdir(), f"pfa_modes_{os.getpid()}.csv")
    for label, extra in (("plain", []), ("--cds --jc", ["--cds", "--jc"]), ("--pops", ["-p", "pop1,pop2"]),
                         ("--pops --cds --jc", ["-p", "pop1,pop2", "--cds", "--jc"]), ("--sfs", ["--sfs"])):
        best = None
        for rep in range(3):
            if os.path.exists(out):
                os.unlink(out)
            t0 = time.perf_counter()
            with redirect_stdout(io.StringIO()):
                cli.main(["-d", d, "-o", out, "-s"] + extra)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rows = sum(1 for _ in open(out))
        print(f"{label:20s} {M / best:9.0f} alignments/s  {rows} rows  {best * 1e3:7.1f} ms", flush=True)
    os.unlink(out)
finally:
    shutil.rmtree(d, ignore_errors=True)
