#!/bin/bash
# Runs started by a SHELL (no threads, no GPU context in the parent): wall time by date +%s%N around the command; one file, and a
# directory of 16 384 files (made first); OPENBLAS_NUM_THREADS=1 (what __main__ sets) against 256 (numpy's default on these boxes),
# with and without the sleeping threads at exit (PFA_EXIT_KICK); alternating rounds.
R=${GRAFT_REPO_ROOT:-$(pwd)}
F=$R/tests/golden/example_theta_0.01/file1.fa
D=/dev/shm/pfa_shell_cold_$$
cd $R
python3 - <<PY
import ctypes, os, sys
sys.path.insert(0, "$R")
import numpy as np
from polyfasta_amd import _lib, scan, synth
lib = _lib.load(); N, L, M = 50, 10000, 16384
os.makedirs("$D"); p = synth.make_params(N, 0.01); stride = scan.row_stride_for(L, 128)
hdr = [f">{h}\n".encode() for h in synth.headers_for(N)]
for m0 in range(0, M, 2048):
    sp = _lib.SynthParamsC(seed=7, first_aln=m0, n_aln=2048, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh, tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
    bases = np.zeros(2048 * N * stride, dtype=np.uint8)
    _lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)), _lib.ptr(np.ascontiguousarray(p.units))))
    for m in range(2048):
        mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
        open(os.path.join("$D", f"aln{m0 + m:06d}.fa"), "wb").write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
PY
for round in 1 2 3; do
  for v in "1 32" "1 0" "256 0"; do
    set -- $v
    l1="round $round one file      OPENBLAS_NUM_THREADS=$1 PFA_EXIT_KICK=$2:"
    l2="round $round 16384 files   OPENBLAS_NUM_THREADS=$1 PFA_EXIT_KICK=$2:"
    for i in 1 2 3 4 5 6; do
      t0=$(date +%s%N); OPENBLAS_NUM_THREADS=$1 PFA_EXIT_KICK=$2 python3 -m polyfasta_amd -f $F > /dev/null 2>&1; t1=$(date +%s%N)
      l1="$l1 $(( (t1 - t0) / 1000000 ))"
    done
    for i in 1 2 3 4; do
      t0=$(date +%s%N); OPENBLAS_NUM_THREADS=$1 PFA_EXIT_KICK=$2 python3 -m polyfasta_amd -d $D -o /tmp/pfa_shell_cold_$$.csv -s > /dev/null 2>&1; t1=$(date +%s%N)
      l2="$l2 $(( (t1 - t0) / 1000000 ))"
    done
    echo "$l1 ms"; echo "$l2 ms"
  done
done
rm -rf $D /tmp/pfa_shell_cold_$$.csv
