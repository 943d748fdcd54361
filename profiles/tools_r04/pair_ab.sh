#!/bin/bash
# scan_kernel with two wavefronts per tile buffer (PFA_PAIR=1): parity first (tests + soak with the pairs on), then the A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1; mkdir -p $O
PFA_PAIR=1 timeout -k 10 500 python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_random.py $R/tests/test_many_rows_gpu.py $R/tests/test_buckets_gpu.py $R/tests/test_baseline_configs_gpu.py $R/tests/test_cli_gpu.py -x -q --deselect tests/test_buckets_gpu.py::test_mixed_batch_costs_the_sum_of_its_parts > $O/tests_pair.log 2>&1 || { tail -30 $O/tests_pair.log; exit 1; }
tail -2 $O/tests_pair.log
PFA_PAIR=1 timeout -k 10 300 python3 $R/tools/soak_gpu.py ${SOAK:-600} 3000 > $O/soak_pair.log 2>&1 || { tail -30 $O/soak_pair.log; exit 1; }
tail -1 $O/soak_pair.log
ROUNDS=${ROUNDS:-3} $R/tools/ab_env.sh $O/ab_pair.txt "${WL:-c4 c4:8192}" "" "PFA_PAIR=1" > /dev/null
cat $O/ab_pair.txt
