#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "OMP_NUM_THREADS=1" "OMP_NUM_THREADS=256" "OMP_NUM_THREADS=1 MKL_NUM_THREADS=256" ; do
    echo "== round $round OPENBLAS_NUM_THREADS=1 $v, 3 files"
    python3 $R/tools/cold_dir_probe.py 3 4 OPENBLAS_NUM_THREADS=1 $v 2>&1 | grep -E "^run [123]|exit  |pfa_init" | awk '/^run/{w=$6} /pfa_init/{i=$(NF-1)} /exit  /{printf "   wall %s ms, pfa_init %s ms, exit %s ms\n", w, i, $(NF-1)}'
  done
done
