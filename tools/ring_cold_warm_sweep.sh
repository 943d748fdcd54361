#!/bin/bash
# The pinned ring (slots x slot bytes; the device text ring is twice the slots): whole-process time of a 16 384-file job in a
# FRESH process (tools/cold_dir_probe.py: wall, files -> rows, exit) next to the warm in-process rate (tools/e2e_trace.py);
# ROUNDS alternating rounds.  usage: tools/ring_cold_warm_sweep.sh OUTFILE ["slots MiB" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
CFGS=("$@"); [ ${#CFGS[@]} = 0 ] && CFGS=("8 16" "4 16" "8 8" "4 8" "4 4" "2 8")
: > $OUT
for round in $(seq 1 ${ROUNDS:-3}); do
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  export PFA_INGEST_SLOTS=$1 PFA_INGEST_SLOT_BYTES=$(( $2 << 20 ))
  echo "== round $round: slots $1 x $2 MiB (pinned $(( $1 * $2 )) MiB, device $(( 2 * $1 * $2 )) MiB)" >> $OUT
  python3 $R/tools/cold_dir_probe.py 16384 5 2>&1 | grep -E "^run [1234]|files -> rows|exit  " | awk '/^run/{w=$6} /files -> rows/{f=$(NF-1)} /exit  /{printf "   fresh process: wall %s ms, files->rows %s ms, exit %s ms\n", w, f, $(NF-1)}' | tail -4 >> $OUT
  PFA_REPS=9 python3 $R/tools/e2e_trace.py 16384 2>&1 | tail -1 | sed 's/^/   in process:   /' >> $OUT
done
done
cat $OUT
