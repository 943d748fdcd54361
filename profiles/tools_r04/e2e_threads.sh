R=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2; do
for t in 16 20 24 32; do
  export PFA_THREADS=$t
  python3 $R/bench.py --no-cpu-baseline --legs end_to_end --steps 5 --warmup 1 2>/dev/null | python3 -c '
import sys,json,os
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=j["end_to_end"]
print("threads=%s " % os.environ.get("PFA_THREADS"), "  ".join("%s %.0fk (med %.0fk)" % (m, e[m]["best"]/1e3, e[m]["median"]/1e3) for m in ("plain","cds_jc","pops","pops_cds_jc")), " read %.4f pack %.4f wait %.4f emit %.4f" % tuple(e["plain"]["stages"][k] for k in ("read_thr","pack_thr","slot_wait_thr","rows")))'
done; done
