#!/bin/bash
# Coding units by row count: 128-site tiles (default above 64 rows) against 256-site tiles / the row-chunked kernel.
# usage: tools/cds_rows_sweep.sh [tag] -> gpurun_out/cds_rows_<tag>.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cds_rows_${1:-x}.txt
: > $OUT
one() {
  local label=$1; shift
  local envs=()
  while [[ "$1" == *=* ]]; do envs+=("$1"); shift; done
  local line
  line=$(env "${envs[@]}" python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 "$@" 2>/dev/null)
  echo "$label ${envs[*]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms  lds %d  %s" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"], j["config"]["lds_bytes_per_wave"], j["roofline"]["kernel"][:52]))')" >> $OUT
}
for n in 80 100 128 160 200 250; do
  b=$((1638400 / n))
  one cds$n X=1 --rows $n --sites 9000 --coding --batch $b
  one cds$n PFA_FORCE_TW=256 --rows $n --sites 9000 --coding --batch $b
done
cat $OUT
