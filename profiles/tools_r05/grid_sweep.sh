#!/bin/bash
# workgroups of scan_kernel's persistent launch against throughput (PFA_GRID overrides "one per wavefront slot").
# usage: tools/grid_sweep.sh OUTFILE
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1
PFA_TRACE_LAUNCH=1 python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 --rows 200 --sites 100000 --batch 128 2>&1 | grep PFA_TRACE_LAUNCH | head -2 > $O.trace
PFA_TRACE_LAUNCH=1 python3 $R/bench.py --no-cpu-baseline --no-extras --steps 2 --warmup 1 2>&1 | grep PFA_TRACE_LAUNCH | head -2 >> $O.trace
ROUNDS=1 $R/tools/ab_env.sh $O.n200 "n200 pops" "" "PFA_GRID=768" "PFA_GRID=1024" "PFA_GRID=1280" "PFA_GRID=1536" "PFA_GRID=2048" "PFA_GRID=3072" "PFA_GRID=6144" "PFA_GRID=30000" > /dev/null
ROUNDS=1 $R/tools/ab_env.sh $O.headline "headline" "" "PFA_GRID=2048" "PFA_GRID=2560" "PFA_GRID=3072" "PFA_GRID=4096" "PFA_GRID=6144" "PFA_GRID=40000" > /dev/null
cat $O.trace $O.n200 $O.headline
