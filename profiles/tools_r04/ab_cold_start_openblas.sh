#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "OPENBLAS_NUM_THREADS=1" "OPENBLAS_NUM_THREADS=256"; do
    env $v python3 $R/bench.py --no-cpu-baseline --legs cold_start --steps 5 --warmup 1 2>/dev/null | python3 -c "
import sys,json
j=json.loads(sys.stdin.read().strip().splitlines()[-1])['cold_start']
print('round $round $v ::', {k:(round(v['median_s'],3), round(v['min_s'],3), [round(x,3) for x in v['phases_s']]) for k,v in j.items() if isinstance(v,dict)})"
  done
done
