#!/bin/bash
# Timing sweep of the row-chunked kernel (scan_tall_kernel) on the GPU box.
# usage: tools/tall_sweep.sh   (writes gpurun_out/tall_sweep.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tall_sweep.txt
: > $OUT
run() {  # label, env..., -- bench args
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local line
  line=$(env "${envs[@]}" python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null)
  echo "$label ${envs[*]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  kernel %.3f ms  lds %d  rpc %s" % (j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["config"]["lds_bytes_per_wave"], j["config"].get("rows_per_chunk")))')" >> $OUT
}
# config 3 shape: 128 matrices of 200 x 100000, two interleaved populations of 100 rows
for rows in 64 52 48 36 28; do for slots in 48 96; do
  run c3_pops PFA_TALL_ROWS=$rows PFA_TALL_SLOTS=$slots -- --rows 100 --pops 2 --sites 100000 --batch 128
done; done
run c3_pops_classic PFA_TALL_MIN=100000 -- --rows 100 --pops 2 --sites 100000 --batch 128
# N = 200 without populations
for rows in 64 52 40 36; do
  run n200 PFA_TALL_ROWS=$rows -- --rows 200 --sites 100000 --batch 128
done
run n200_classic PFA_TALL_MIN=100000 -- --rows 200 --sites 100000 --batch 128
# the headline shape through the row-chunked kernel
for rows in 52 28; do
  run n50_tall PFA_TALL_MIN=0 PFA_TALL_ROWS=$rows -- 
done
run n50_classic X=1 --
run n1000 X=1 -- --rows 1000 --sites 20000 --batch 128
run n1000_classic PFA_TALL_MIN=100000 -- --rows 1000 --sites 20000 --batch 64
cat $OUT
