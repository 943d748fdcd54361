#!/bin/bash
# Where does the row-chunked kernel overtake the all-rows-in-LDS kernel, and with how many slots?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tall_sweep2.txt
: > $OUT
run() {
  local label=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  local line
  line=$(env "${envs[@]}" python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null)
  echo "$label ${envs[*]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  kernel %.3f ms  lds %d  rpc %s" % (j["roofline"]["achieved"], j["roofline"]["kernel_ms_avg"], j["config"]["lds_bytes_per_wave"], j["config"].get("rows_per_chunk")))')" >> $OUT
}
for n in 72 80 100 128 160 200 400; do
  b=$((25600 / n))
  run "N=$n classic" PFA_TALL_MIN=100000 -- --rows $n --sites 100000 --batch $b
  for slots in 32 48 64 96; do
    run "N=$n tall" PFA_TALL_SLOTS=$slots -- --rows $n --sites 100000 --batch $b
  done
done
cat $OUT
