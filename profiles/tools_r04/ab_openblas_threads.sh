#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "OPENBLAS_NUM_THREADS=1" "OPENBLAS_NUM_THREADS=256"; do
    echo "== round $round $v, 3 files"
    python3 $R/tools/cold_dir_probe.py 3 4 $v OMP_NUM_THREADS=256 2>&1 | grep -E "^run [123]|exit  |pfa_init|import cli"
    echo "== round $round $v, 16384 files"
    python3 $R/tools/cold_dir_probe.py 16384 4 $v OMP_NUM_THREADS=256 2>&1 | grep -E "^run [123]|exit  |pfa_init|files -> rows"
  done
done
