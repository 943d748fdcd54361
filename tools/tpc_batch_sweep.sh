#!/bin/bash
# tiles per chunk x batch size for the 128-site-tile shapes: is the loss per tile (processing) or per workgroup / per launch
# (prologue latency, the last round of workgroups)?  usage: tools/tpc_batch_sweep.sh OUTFILE
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
: > $OUT
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
one() {
  local label=$1; shift
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 3 "$@" 2>>$OUT.err)
  echo "$label :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("frac %.3f  %.0f GB/s  kernel %.3f ms (min %.3f)" % (j["roofline"]["frac"], j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["timing_ms"]["kernel_min"]))')" >> $OUT
}
for rep in 1 2; do
for tpc in default 2 4 8 16 32 64 128; do
  if [ $tpc = default ]; then unset PFA_TILES_PER_CHUNK; else export PFA_TILES_PER_CHUNK=$tpc; fi
  one "rep $rep n200 batch 128 tpc $tpc" --rows 200 --sites 100000 --batch 128
done
unset PFA_TILES_PER_CHUNK
for b in 128 256 512 1024 2048; do
  one "rep $rep n200 batch $b tpc default" --rows 200 --sites 100000 --batch $b
done
for b in 256 1024 4096; do
  one "rep $rep n100 batch $b tpc default" --rows 100 --sites 100000 --batch $b
done
for tpc in 8 32 128; do
  PFA_TILES_PER_CHUNK=$tpc one "rep $rep n200 batch 1024 tpc $tpc" --rows 200 --sites 100000 --batch 1024
done
for tpc in 12 24 48 72; do
  PFA_TILES_PER_CHUNK=$tpc one "rep $rep n200cds batch 8192 tpc $tpc" --rows 200 --sites 9000 --coding --batch 8192
done
done
cat $OUT
