#!/bin/bash
# the classic kernel's workloads (kernel GB/s from HIP events), for this build and, when given, other ones side by side.
# usage: tools/bench_classic.sh [tag] [other libpfa_hip .so ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/classic_${1:-x}.txt
shift
OTHER="$*"
: > $OUT
one() {
  local label=$1; shift
  for lib in "" $OTHER; do
    line=$(PFA_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 3 "$@" 2>>$OUT.err)
    echo "$label ${lib:+[$(basename $lib)]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms (min %.3f)  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:60]))')" >> $OUT
  done
}
# the first seconds of a box run at higher clocks (the first measurement of a call reads 2 % high): spend them on a discarded run
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2000 --warmup 3 > /dev/null 2>&1
one headline
one c4 --coding --sites 9000
one pops --rows 100 --pops 2 --sites 100000 --batch 128
one n100 --rows 100 --sites 100000 --batch 256
one n200 --rows 200 --sites 100000 --batch 128
one n200cds --rows 200 --sites 9000 --coding --batch 8192
one n100cds --rows 100 --sites 9000 --coding --batch 16384
cat $OUT
