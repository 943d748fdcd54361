#!/bin/bash
# The driver's 8-GPU shape on a one-GPU box, as far as the box allows: a GPU box admits at most SIX processes on its card, so the
# eight ranks of `bench.py --gpus 8` cannot share it.  What eight ranks change against six is each rank's share of the control
# group's 16 CPUs — 2 instead of 2.67 — and that can be had with six: PFA_CPUS=2 gives every rank the loader pool (2 threads) and
# the wait policy (sleeping) it would take as one of eight; PFA_CPUS=1 is one step further (one loader per rank).
# usage: tools/rehearse_cpu_share.sh OUTDIR [TOTAL_FILES]
set -u
OUT=${1:-gpurun_out/rehearsal_cpu_share}; TOTAL=${2:-12288}
mkdir -p "$OUT"
R0=$(cd "$(dirname "$0")/.." && pwd)
{
  echo "nproc: $(nproc)   cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  python3 -c "import os; print('sched_getaffinity:', len(os.sched_getaffinity(0)))"
} > "$OUT/box.txt" 2>&1
run() {  # name, ranks, extra env...
  local name=$1 r=$2; shift 2
  local per=$(( TOTAL / r ))
  env PFA_ALLOW_SHARED_GPU=1 PFA_COMM_FACTORY=tests._gloo_comm:GlooComm PYTHONPATH=$R0 "$@" \
    timeout -k 10 400 python3 $R0/bench.py --gpus $r --steps 3 --warmup 2 --batch 512 --e2e-files-per-rank $per --e2e-reps 4 \
    > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$name rc=$?" >> "$OUT/rc.txt"
  python3 - "$OUT/$name.json" "$name" <<'PY' >> "$OUT/summary.txt"
import json, sys
try:
    e = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["end_to_end"]
    print(sys.argv[2], "ranks", e["ranks"], "files", e["files"], "best %.0f median %.0f /s  first %.3f s | solo best %.0f median %.0f | rows_ok %s | cpus_per_rank %s | host_read_ceiling %s" % (
        e["best"], e["median"], e["first_s"], e["solo_rank0_same_files"]["best"], e["solo_rank0_same_files"]["median"], e["rows_ok"],
        e.get("cpus_per_rank"), e.get("host_read_ceiling_alignments_per_s")))
    for k, v in e["stages_per_rank_of_best_run_s"].items():
        print("    %-14s" % k, v)
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run "ranks6_share_of_6" 6
run "ranks6_share_as_one_of_8_PFA_CPUS_2" 6 PFA_CPUS=2
run "ranks6_PFA_CPUS_1" 6 PFA_CPUS=1
cat "$OUT/box.txt" "$OUT/summary.txt"
