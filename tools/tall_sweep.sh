#!/bin/bash
# Row-chunked kernel: rows per chunk x count-table slots, coding and not (kernel GB/s from HIP events).
# usage: tools/tall_sweep.sh [tag] -> gpurun_out/tall_sweep_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tall_sweep_${1:-x}.txt
: > $OUT
one() {
  local label=$1; shift
  local envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  local line
  line=$(env "${envs[@]}" python3 $R/bench.py --no-cpu-baseline --no-extras --steps 12 --warmup 2 "$@" 2>/dev/null)
  echo "$label ${envs[*]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms  lds %d  rpc %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["config"]["lds_bytes_per_wave"], j["config"].get("rows_per_chunk")))')" >> $OUT
}
for rows in 28 32 36 40 44 48 56; do
  for slots in 32 64; do
    one n1000cds PFA_TALL_ROWS=$rows PFA_TALL_SLOTS=$slots --rows 1000 --sites 9000 --coding --batch 2048
  done
done
for rows in 32 40 48; do
  one n5000cds PFA_TALL_ROWS=$rows PFA_TALL_SLOTS=32 --rows 5000 --sites 9000 --coding --batch 1024
  one n1000 PFA_TALL_ROWS=$rows --rows 1000 --sites 100000 --batch 128
done
cat $OUT
