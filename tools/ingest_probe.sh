#!/bin/bash
# host-ingest ceiling probe on the GPU box: tools/ingest_probe.sh [files]
set -e
M=${1:-4096}
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
D=/dev/shm/pfa_probe_$$
python3 - "$M" "$D" <<'P'
import os, sys, ctypes, numpy as np
sys.path.insert(0, ".")
from polyfasta_amd import _lib, scan, synth
M, d = int(sys.argv[1]), sys.argv[2]
lib = _lib.load(); N, L = 50, 10000
p = synth.make_params(N, 0.01); stride = scan.row_stride_for(L, 128)
sp = _lib.SynthParamsC(seed=7, first_aln=0, n_aln=M, n_rows=N, n_sites=L, row_stride=stride, seg_thresh=p.seg_thresh,
                       tri_thresh=p.tri_thresh, n_units=len(p.units), coding=0)
bases = np.zeros(M * N * stride, dtype=np.uint8)
_lib.check(lib.pfa_synth_fill_host(_lib.ptr(bases), ctypes.byref(sp), _lib.ptr(np.ascontiguousarray(p.cdf)), _lib.ptr(np.ascontiguousarray(p.units))))
os.makedirs(d)
hdr = [f">indiv{i + 1}\n".encode() for i in range(N)]
for m in range(M):
    mat = bases[m * N * stride:(m + 1) * N * stride].reshape(N, stride)[:, :L]
    with open(os.path.join(d, f"aln{m:06d}.fa"), "wb") as fh:
        fh.write(b"".join(h + mat[i].tobytes() + b"\n" for i, h in enumerate(hdr)))
print("wrote", M, "files to", d, "cpus", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
P
P=tools/ubench/ingest_probe
$P $D 1 4
for T in 8 16 32 64 128; do $P $D $T 0; done
for T in 8 16 32 64 128; do $P $D $T 1; done
for T in 16 32 64; do $P $D $T 2; done
for T in 16 32 64 128; do $P $D $T 3 32; done
$P $D 64 3 8
$P $D 64 3 128
for T in 16 32 64 128; do echo "== CLI with PFA_THREADS=$T"; python3 tools/e2e_bench.py --files $M --dir $D --threads $T; done
rm -rf $D
