#!/bin/bash
# Profile bench.py on the GPU box: kernel trace + stats, then HBM counters in
# separate passes (MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE do not fit one pass).
# usage: tools/gpu_prof.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-extras "$@" > $OUT/bench_write.json 2> $OUT/write.err
find $OUT -name "*.csv" | head -20
