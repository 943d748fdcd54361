#!/bin/bash
# row stride (bytes between rows of a matrix = round_up(L, --stride-align)) against kernel rate; usage: tools/stride_ab.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/bench.py --no-cpu-baseline --no-extras --steps 1500 --warmup 3 > /dev/null 2>&1
for round in 1 2; do
for w in "--rows 50 --sites 10000" "--rows 1000 --sites 9000 --batch 2048 --coding" "--rows 5000 --sites 9000 --batch 1024 --coding" "--rows 200 --sites 9000 --batch 8192 --coding"; do
for a in 128 256 512 1024 4096; do
  line=$(python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 $w --stride-align $a 2>/dev/null)
  echo "round $round [$w] stride-align $a :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("stride %d  frac %.3f  %.0f GB/s  (min %.3f ms)  %s" % (j["config"]["row_stride"], j["roofline"]["frac"], j["roofline"]["achieved"], j["timing_ms"]["kernel_min"], j["roofline"]["kernel"][:60]))')"
done; done; done
