#!/bin/bash
# round-N profiles of the final sources: rocprofv3 kernel stats + PMC traffic for every roofline workload, and the kernel
# table of one files -> CSV run (repack_kernel beside the scan)
RN=${1:-r5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/gpu_prof.sh ${RN}_head --steps 30 --warmup 3 > /dev/null 2>&1; echo head done
bash tools/gpu_prof.sh ${RN}_coding --steps 30 --warmup 3 --coding --sites 9000 > /dev/null 2>&1; echo coding done
bash tools/gpu_prof.sh ${RN}_pops --steps 30 --warmup 3 --rows 100 --pops 2 --sites 100000 --batch 128 > /dev/null 2>&1; echo pops done
bash tools/gpu_prof.sh ${RN}_n200 --steps 30 --warmup 3 --rows 200 --sites 100000 --batch 128 > /dev/null 2>&1; echo n200 done
bash tools/gpu_prof.sh ${RN}_n200cds --steps 30 --warmup 3 --rows 200 --sites 9000 --coding --batch 8192 > /dev/null 2>&1; echo n200cds done
bash tools/gpu_prof.sh ${RN}_n1000cds --steps 20 --warmup 3 --rows 1000 --sites 9000 --coding --batch 2048 > /dev/null 2>&1; echo n1000cds done
bash tools/gpu_prof.sh ${RN}_n5000cds --steps 12 --warmup 2 --rows 5000 --sites 9000 --coding --batch 1024 > /dev/null 2>&1; echo n5000cds done
mkdir -p $R/gpurun_out/prof_${RN}_e2e
cd /tmp && export TMPDIR=/tmp
PFA_REPS=4 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${RN}_e2e/trace -- python3 $R/tools/e2e_trace.py 8192 > $R/gpurun_out/prof_${RN}_e2e/run.txt 2> $R/gpurun_out/prof_${RN}_e2e/err.txt
echo e2e done
python3 $R/tools/small_launch_probe.py > $R/gpurun_out/${RN}_small_launch_probe.txt 2>&1; echo probe done
