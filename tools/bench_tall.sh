#!/bin/bash
# the row-chunked kernel's workloads (kernel GB/s from HIP events), for this build and, when given, another one side by side.
# usage: tools/bench_tall.sh [tag] [other libpfa_hip .so]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tall_${1:-x}.txt
OTHER="$2 $3 $4"
: > $OUT
one() {
  local label=$1; shift
  for lib in "" $OTHER; do
    line=$(PFA_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 "$@" 2>>$OUT.err)
    echo "$label ${lib:+[$(basename $lib)]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms (min %.3f)  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_ms_min"], j["roofline"]["kernel"][:70]))')" >> $OUT
  done
}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2000 --warmup 3 > /dev/null 2>&1   # (discarded: the first seconds of a box run at higher clocks)
one n1000cds --rows 1000 --sites 9000 --coding --batch 2048
one n5000cds --rows 5000 --sites 9000 --coding --batch 1024
one n1000 --rows 1000 --sites 100000 --batch 128
one n5000 --rows 5000 --sites 100000 --batch 32
one n300cds --rows 300 --sites 9000 --coding --batch 4096
cat $OUT
