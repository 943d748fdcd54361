R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export PFA_NO_AVX512=1; else unset PFA_NO_AVX512; fi
  python3 $R/bench.py --no-cpu-baseline --legs end_to_end --steps 5 --warmup 1 2>/dev/null | python3 -c '
import sys,json,os
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j["end_to_end"]
print("no_avx512=%s " % os.environ.get("PFA_NO_AVX512","0"), "  ".join("%s %.0fk (med %.0fk)" % (m, e[m]["best"]/1e3, e[m]["median"]/1e3) for m in ("plain","cds_jc","pops","pops_cds_jc")), " pack s/thread %.4f" % e["plain"]["stages"]["pack_thr"])'
done; done
