#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
for round in 1 2; do
for w in "--rows 1000 --sites 9000 --batch 2048" "--rows 1000 --sites 9000 --batch 2048 --coding" "--rows 200 --sites 100000 --batch 128" "--rows 50 --sites 10000"; do
for v in 0 2; do
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 24 --warmup 3 $w --variant $v 2>/dev/null)
  echo "round $round [$w] variant $v :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("frac %.3f  %.0f GB/s  (min %.3f ms)  %s" % (j["roofline"]["frac"], j["roofline"]["achieved"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:70]))')"
done; done; done
