R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tpc_sweep.txt
: > $OUT
one() {
  local label=$1; shift
  for tpc in 0 1 2 3 6; do
    if [ $tpc = 0 ]; then unset PFA_TILES_PER_CHUNK; else export PFA_TILES_PER_CHUNK=$tpc; fi
    line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 "$@" 2>>$OUT.err)
    echo "$label tpc=$tpc :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["roofline"]["kernel"][-40:]))')" >> $OUT
  done
}
one n1000 --rows 1000 --sites 100000 --batch 128
one n1000cds --rows 1000 --sites 9000 --coding --batch 2048
one n300 --rows 300 --sites 100000 --batch 128
one n300cds --rows 300 --sites 9000 --coding --batch 4096
cat $OUT
