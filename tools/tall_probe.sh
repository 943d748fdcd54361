R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tall_probe.txt
: > $OUT
one() {
  local label=$1; shift
  for lib in "" polyfasta_amd/libpfa_hip_p2.so polyfasta_amd/libpfa_hip_p3.so polyfasta_amd/libpfa_hip_p4.so; do
    line=$(PFA_LIB=$lib python3 $R/bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 3 "$@" 2>>$OUT.err)
    echo "$label ${lib:+[$(basename $lib)]} :: $(echo "$line" | python3 -c 'import sys,json; j=json.loads(sys.stdin.read()); print("%.0f GB/s  frac %.3f  kernel %.3f ms" % (j["roofline"]["achieved"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"]))')" >> $OUT
  done
}
one n1000cds --rows 1000 --sites 9000 --coding --batch 2048
one n5000cds --rows 5000 --sites 9000 --coding --batch 1024
one n300cds --rows 300 --sites 9000 --coding --batch 4096
cat $OUT
