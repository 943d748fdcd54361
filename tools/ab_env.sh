#!/bin/bash
# A/B of environment settings on the scan workloads (kernel GB/s from HIP events): every workload with every setting, `ROUNDS`
# alternating rounds, a discarded run first.  usage: tools/ab_env.sh OUTFILE "workload[:batch] ..." "VAR=x VAR2=y" ["..."]   ("" = none)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; WL=$2; shift 2
SETS=("$@")
: > $OUT
args_of() {
  case $1 in
    headline) echo "--rows 50 --sites 10000" ;;
    c4) echo "--coding --rows 50 --sites 9000" ;;
    pops) echo "--rows 100 --pops 2 --sites 100000" ;;
    n100) echo "--rows 100 --sites 100000" ;;
    n144) echo "--rows 144 --sites 100000" ;;
    n200) echo "--rows 200 --sites 100000" ;;
    n200cds) echo "--rows 200 --sites 9000 --coding" ;;
    n100cds) echo "--rows 100 --sites 9000 --coding" ;;
    n1000cds) echo "--rows 1000 --sites 9000 --coding" ;;
    n5000cds) echo "--rows 5000 --sites 9000 --coding" ;;
    n1000) echo "--rows 1000 --sites 9000" ;;
    n300cds) echo "--rows 300 --sites 9000 --coding" ;;
  esac
}
batch_of() {
  case $1 in headline|c4) echo 32768 ;; pops|n200) echo 128 ;; n100) echo 256 ;; n144) echo 177 ;; n200cds) echo 8192 ;; n100cds) echo 16384 ;;
    n1000cds|n1000) echo 2048 ;; n5000cds) echo 1024 ;; n300cds) echo 4096 ;; esac
}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
for round in $(seq 1 ${ROUNDS:-2}); do
  for wb in $WL; do
    w=${wb%%:*}; b=${wb#*:}; [ "$b" = "$wb" ] && b=$(batch_of $w)
    for set in "${SETS[@]}"; do
      line=$(env $set python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 3 $(args_of $w) --batch $b 2>>$OUT.err)
      echo "round $round $w batch $b [$set] :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("frac %.3f  %.0f GB/s  kernel %.3f ms (min %.3f)  %s" % (j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:70]))')" >> $OUT
    done
  done
done
cat $OUT
