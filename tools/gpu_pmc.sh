#!/bin/bash
# SQ activity counters of the scan kernels (separate --pmc passes, no tracing): where do the wave cycles go?
# usage: tools/gpu_pmc.sh <tag> [bench args...]     -> gpurun_out/pmc_<tag>.txt (+ raw CSVs in gpurun_out/pmc_<tag>/)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU"
      "SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
      "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL")
i=0
for S in "${SETS[@]}"; do
  rocprofv3 --pmc $S --output-format csv -d $OUT/sq$i -- python3 $R/bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 2 "$@" > $OUT/bench$i.json 2> $OUT/err$i.txt
  i=$((i+1))
done
python3 - <<PY > $R/gpurun_out/pmc_$TAG.txt
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/sq*/*/*_counter_collection.csv"):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "scan_" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    w = sum(per["SQ_WAVE_CYCLES"]) / max(len(per["SQ_WAVE_CYCLES"]), 1)
    for k, v in per.items():
        agg[k] = (sum(v) / len(v), w)
for k, (m, w) in sorted(agg.items()):
    print(f"$TAG {k:26s} {m:18.0f}  {m / w if w else 0:8.4f} of wave cycles")
PY
cat $R/gpurun_out/pmc_$TAG.txt
