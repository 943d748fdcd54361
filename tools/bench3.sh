#!/bin/bash
# The three roofline workloads of DESIGN.md §3 in one go (kernel GB/s from HIP events).
# usage: tools/bench3.sh [tag]   -> gpurun_out/bench3_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bench3_${1:-x}.txt
: > $OUT
one() {
  local label=$1; shift
  local line
  line=$(python3 $R/bench.py --no-cpu-baseline --steps 30 --warmup 3 "$@" 2>/dev/null)
  echo "$label :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  kernel %.3f ms (min %.3f)  lds %d  value %.3g" % (j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_ms_min"], j["config"]["lds_bytes_per_wave"], j["value"]))')" >> $OUT
}
one headline
one coding --coding --sites 9000
one pops --rows 100 --pops 2 --sites 100000 --batch 128
one n200 --rows 200 --sites 100000 --batch 128
cat $OUT
