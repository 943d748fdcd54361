#!/bin/bash
# round-N final measurements of the final sources in one call: rocprofv3 stats + PMC traffic per workload (tools/prof_all.sh $RN),
# SQ counters, the default bench line (+ its long form), the multi-rank rehearsals
RN=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/prof_all.sh $RN > gpurun_out/${RN}_final_prof.log 2>&1
for w in "head" "coding --coding --sites 9000" "pops --rows 100 --pops 2 --sites 100000 --batch 128" "n200 --rows 200 --sites 100000 --batch 128" \
         "n200cds --rows 200 --sites 9000 --coding --batch 8192" "n1000cds --rows 1000 --sites 9000 --coding --batch 2048" "n5000cds --rows 5000 --sites 9000 --coding --batch 1024"; do
  set -- $w; tag=$1; shift
  bash tools/gpu_pmc.sh ${RN}_$tag "$@" > /dev/null 2>&1; echo "pmc $tag done"
done
python3 bench.py > gpurun_out/${RN}_final_bench_n1.json 2> gpurun_out/${RN}_final_bench_n1.err; echo "bench rc=$?"
bash tools/rehearse_cpu_share.sh gpurun_out/${RN}_final_rehearsal 12288 > gpurun_out/${RN}_final_rehearsal.log 2>&1; echo "rehearsal done"
