#!/bin/bash
# Rehearse the multi-process `--dir` path (RCCL process group, record all-gather, rank-0 emit)
# with ONE rank on the one-GPU box, and compare its CSV with the single-process run.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d /dev/shm/pfa_dist_XXXX)
python3 - "$D" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from polyfasta_amd import synth
d = sys.argv[1]
p = synth.make_params(20, 0.02)
for k in range(300):
    m = synth.synth_alignment(4242, k, 3000 + 3 * (k % 7), p, coding=True)
    open(os.path.join(d, f"a{k:04d}.fa"), "w").write(synth.to_fasta_text(synth.headers_for(20, 2), m))
open(os.path.join(d, "zz_notfasta.fa"), "w").write("hello\n")
PY
cd $R
python3 -m polyfasta_amd -d $D -o $D/../single.csv -s
python3 -m polyfasta_amd -d $D -o $D/../single_cds.csv -s --cds --jc -p pop1,pop2 > $D/../single_cds.stdout
PFA_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 -m polyfasta_amd -d $D -o $D/../dist.csv -s 2> $D/../dist.err
PFA_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29545 -m polyfasta_amd -d $D -o $D/../dist_cds.csv -s --cds --jc -p pop1,pop2 > $D/../dist_cds.stdout 2>> $D/../dist.err
cmp $D/../single.csv $D/../dist.csv && cmp $D/../single_cds.csv $D/../dist_cds.csv && cmp $D/../single_cds.stdout $D/../dist_cds.stdout && echo "dist rehearsal: identical CSV ($(wc -l < $D/../dist.csv) + $(wc -l < $D/../dist_cds.csv) rows)"
rm -rf $D $D/../single*.csv $D/../dist*.csv
