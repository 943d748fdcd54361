#!/bin/bash
# bench.py --gpus R as R ranks SHARING the one GPU of a test box (gloo injected: RCCL refuses two ranks per device): the N > 1
# end-to-end leg with its per-rank stage table, in three settings per rank count — as shipped (loader threads and wait policy
# from the rank's share of the control group's CPU quota), the same without the CPU binding (PFA_NO_AFFINITY=1), and as round 3
# ran it (16 loaders per rank, spinning waits).
# usage: tools/rehearse_ranks.sh OUTDIR "2 4 6" TOTAL_FILES
set -u
OUT=${1:-gpurun_out/rehearsal}; RANKS=${2:-"2 4 6"}; TOTAL=${3:-16384}
mkdir -p "$OUT"
R0=$(cd "$(dirname "$0")/.." && pwd)
{
  echo "nproc: $(nproc)   online: $(cat /sys/devices/system/cpu/online 2>/dev/null)"
  echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)   cpuset: $(cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null)"
  python3 -c "import os; print('sched_getaffinity:', len(os.sched_getaffinity(0)))"
  grep -E "nr_periods|nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
  df -h /dev/shm | tail -1
} > "$OUT/box.txt" 2>&1
run() {  # name, ranks, extra env...
  local name=$1 r=$2; shift 2
  local per=$(( TOTAL / r ))
  grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ' > "$OUT/$name.throttle_before.txt"
  env PFA_ALLOW_SHARED_GPU=1 PFA_COMM_FACTORY=tests._gloo_comm:GlooComm PYTHONPATH=$R0 "$@" \
    timeout -k 10 400 python3 $R0/bench.py --gpus $r --steps 3 --warmup 2 --batch 512 --e2e-files-per-rank $per --e2e-reps 4 \
    > "$OUT/$name.json" 2> "$OUT/$name.err"
  echo "$name rc=$?" >> "$OUT/rc.txt"
  grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ' > "$OUT/$name.throttle_after.txt"
  python3 - "$OUT/$name.json" "$name" <<'PY' >> "$OUT/summary.txt"
import json, sys
try:
    e = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["end_to_end"]
    print(sys.argv[2], "ranks", e["ranks"], "files", e["files"], "best %.0f median %.0f /s  first %.3f s | solo best %.0f median %.0f | beyond solo %.4f s | rows_ok %s" % (
        e["best"], e["median"], e["first_s"], e["solo_rank0_same_files"]["best"], e["solo_rank0_same_files"]["median"], e["s_beyond_solo"], e["rows_ok"]))
    for k, v in e["stages_per_rank_of_best_run_s"].items():
        print("    %-14s" % k, v)
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
for r in $RANKS; do
  run "ranks${r}_auto" $r                                        # as shipped: loaders and wait policy from the CPU share of a rank
  run "ranks${r}_auto_no_affinity" $r PFA_NO_AFFINITY=1
  run "ranks${r}_as_round3_16_loaders_spin" $r PFA_THREADS=16 PFA_WAIT=spin
done
cat "$OUT/box.txt" "$OUT/summary.txt"
