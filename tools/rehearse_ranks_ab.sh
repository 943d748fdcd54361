#!/bin/bash
# the N > 1 end-to-end leg as R ranks sharing the one GPU of a test box (tools/rehearse_ranks.sh), settings side by side in alternating rounds
# usage: tools/rehearse_ranks_ab.sh OUTFILE RANKS ROUNDS "VAR=x ..." ["..."]        ("" = as shipped)
set -u
OUT=$1; R=$2; ROUNDS=$3; shift 3
SETS=("$@")
R0=$(cd "$(dirname "$0")/.." && pwd)
per=$(( 16384 / R ))
: > $OUT
for round in $(seq 1 $ROUNDS); do
  for set in "${SETS[@]}"; do
    line=$(env PFA_ALLOW_SHARED_GPU=1 PFA_COMM_FACTORY=tests._gloo_comm:GlooComm PYTHONPATH=$R0 $set \
      timeout -k 10 400 python3 $R0/bench.py --gpus $R --steps 3 --warmup 2 --batch 512 --e2e-files-per-rank $per --e2e-reps 4 2>>$OUT.err)
    echo "round $round ranks $R [$set] :: $(echo "$line" | python3 -c '
import sys, json
e = json.loads(sys.stdin.read().strip().splitlines()[-1])["end_to_end"]
st = e["stages_per_rank_of_best_run_s"]
print("best %.0f median %.0f /s | solo best %.0f | threads %s | read_thr %s" % (e["best"], e["median"], e["solo_rank0_same_files"]["best"], st["threads"], [round(x, 3) for x in st["read_thr"]]))')" >> $OUT
  done
done
cat $OUT
