#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
: > $OUT
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
one() {
  local label=$1; shift
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 30 --warmup 3 "$@" 2>>$OUT.err)
  echo "$label :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("frac %.3f  %.0f GB/s  kernel %.3f ms (min %.3f) p50 %.3f" % (j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"], j["timing_ms"]["kernel_p50"]))')" >> $OUT
}
for rep in 1 2 3; do
for tpc in default 4 5 8 10 14 20 40; do
  if [ $tpc = default ]; then unset PFA_TILES_PER_CHUNK; else export PFA_TILES_PER_CHUNK=$tpc; fi
  one "rep $rep headline tpc $tpc"
done
for tpc in default 6 9 12 18 36; do
  if [ $tpc = default ]; then unset PFA_TILES_PER_CHUNK; else export PFA_TILES_PER_CHUNK=$tpc; fi
  one "rep $rep c4 tpc $tpc" --coding --sites 9000
done
unset PFA_TILES_PER_CHUNK
done
cat $OUT
