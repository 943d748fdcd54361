R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/e2e_16_vs_20.txt
: > $OUT
for i in 1 2 3 4 5 6; do
  for t in 16 20 24; do
    echo -n "threads=$t " >> $OUT
    PFA_BIND=1 PFA_REPS=9 PFA_THREADS=$t python3 $R/tools/e2e_trace.py 16384 2>&1 | grep "warm best" >> $OUT
  done
done
cat $OUT
