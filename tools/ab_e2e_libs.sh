#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3 4 5 6; do
  for lib in ${ORDER:-"" $R/gpurun_lib/libpfa_ingest_before.so}; do
    echo "round $round [${lib:+before}] $(PFA_LIB=$lib PFA_REPS=9 python3 $R/tools/e2e_trace.py 16384 2>&1 | tail -1)"
  done
done
