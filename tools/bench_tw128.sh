#!/bin/bash
# the non-coding 128-site-tile workloads (kernel GB/s from HIP events).  usage: tools/bench_tw128.sh [tag]   (PFA_LIB selects another build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tw128_${1:-x}.txt
: > $OUT
one() {
  local label=$1; shift
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 "$@" 2>>$OUT.err)
  echo "$label :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms (min %.3f)  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:60]))')" >> $OUT
}
one pops --rows 100 --pops 2 --sites 100000 --batch 128
one n100 --rows 100 --sites 100000 --batch 256
one n144 --rows 144 --sites 100000 --batch 177
one n200 --rows 200 --sites 100000 --batch 128
cat $OUT
