#!/bin/bash
# SQ activity counters of the scan kernel (one --pmc pass, no tracing): where do the wave cycles go?
# usage: tools/gpu_pmc.sh <tag> [bench args...]
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/err.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/sq/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "scan_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
w = sum(agg["SQ_WAVE_CYCLES"]) / len(agg["SQ_WAVE_CYCLES"])
for k, v in sorted(agg.items()):
    m = sum(v) / len(v)
    print(f"$TAG {k:24s} {m:16.0f}  {m / w:6.3f} of wave cycles")
PY
