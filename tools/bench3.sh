#!/bin/bash
# The roofline workloads of DESIGN.md §3 in one go (kernel GB/s from HIP events).
# usage: tools/bench3.sh [tag]   -> gpurun_out/bench3_<tag>.txt     (PFA_LIB selects another build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/bench3_${1:-x}.txt
: > $OUT
one() {
  local label=$1; shift
  local line
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 "$@" 2>>$OUT.err)
  echo "$label :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms (min %.3f)  lds %d  value %.3g  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_ms_min"], j["config"]["lds_bytes_per_wave"], j["value"], j["roofline"]["kernel"][:60]))')" >> $OUT
}
one headline
one coding --coding --sites 9000
one pops --rows 100 --pops 2 --sites 100000 --batch 128
one n200 --rows 200 --sites 100000 --batch 128
one n200cds --rows 200 --sites 9000 --coding --batch 8192
cat $OUT
