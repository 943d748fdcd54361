#!/bin/bash
# A/B of library builds on the scan workloads (kernel GB/s from HIP events): every workload with every build, `ROUNDS` alternating
# rounds (the first seconds of a box run at higher clocks: a discarded run first).
# usage: tools/ab_libs.sh OUTFILE "workload names" lib1.so [lib2.so ...]    ("" = the in-tree build)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; WL=$2; shift 2
LIBS=("" "$@")
: > $OUT
args_of() {
  case $1 in
    headline) echo "" ;;
    c4) echo "--coding --sites 9000" ;;
    pops) echo "--rows 100 --pops 2 --sites 100000 --batch 128" ;;
    n100) echo "--rows 100 --sites 100000 --batch 256" ;;
    n144) echo "--rows 144 --sites 100000 --batch 177" ;;
    n200) echo "--rows 200 --sites 100000 --batch 128" ;;
    n200cds) echo "--rows 200 --sites 9000 --coding --batch 8192" ;;
    n100cds) echo "--rows 100 --sites 9000 --coding --batch 16384" ;;
    n1000cds) echo "--rows 1000 --sites 9000 --coding --batch 2048" ;;
    n5000cds) echo "--rows 5000 --sites 9000 --coding --batch 1024" ;;
    n1000) echo "--rows 1000 --sites 9000 --batch 2048" ;;
    n300cds) echo "--rows 300 --sites 9000 --coding --batch 4096" ;;
  esac
}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
for round in $(seq 1 ${ROUNDS:-2}); do
  for w in $WL; do
    for lib in "${LIBS[@]}"; do
      line=$(PFA_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 3 $(args_of $w) 2>>$OUT.err)
      echo "round $round $w [${lib:+$(basename $lib)}] :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("frac %.3f  %.0f GB/s  kernel %.3f ms (min %.3f)  %s" % (j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:60]))')" >> $OUT
    done
  done
done
cat $OUT
