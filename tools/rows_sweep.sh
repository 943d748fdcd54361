#!/bin/bash
# classic kernel against the row-chunked one by rows per unit (waves per CU of the classic kernel fall with the rows)
# usage: tools/rows_sweep.sh [tag] [extra bench flags, e.g. --coding]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; shift
OUT=$R/gpurun_out/rows_sweep_$TAG.txt
: > $OUT
one() {
  local n=$1; shift
  local b=$(( 25600 / n ))
  local sites=100000
  if [[ "$*" == *--coding* ]]; then sites=9000; b=$(( 1638400 / n )); fi
  line=$(env "${ENVS[@]}" python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 --rows $n --sites $sites --batch $b "$@" 2>/dev/null)
  echo "N=$n ${ENVS[*]} $* :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  lds %d  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["config"]["lds_bytes_per_wave"], j["roofline"]["kernel"][:60]))')" >> $OUT
}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2000 --warmup 3 > /dev/null 2>&1   # (discarded warm-up)
for n in 100 128 144 160 176 192 200 208 224 250 300; do
  ENVS=(PFA_X=1); one $n "$@"
  ENVS=(PFA_TALL_MIN=100); one $n "$@"
done
cat $OUT
