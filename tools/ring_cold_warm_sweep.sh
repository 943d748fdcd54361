#!/bin/bash
# The pinned ring (slots x slot bytes; the device text ring is twice the slots): whole-process time of a 16 384-file job in a
# FRESH process (tools/cold_dir_probe.py: wall, files -> rows, exit) next to the warm in-process rate (tools/e2e_trace.py).
# usage: tools/ring_cold_warm_sweep.sh OUTFILE
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1
: > $OUT
for cfg in "8 16" "4 16" "2 16" "8 4" "4 8" "4 4" "3 8" "2 8" "6 8"; do
  set -- $cfg
  export PFA_INGEST_SLOTS=$1 PFA_INGEST_SLOT_BYTES=$(( $2 << 20 ))
  echo "== slots $1 x $2 MiB (pinned $(( $1 * $2 )) MiB, device $(( 2 * $1 * $2 )) MiB)" >> $OUT
  python3 $R/tools/cold_dir_probe.py 16384 3 2>&1 | grep -E "^run [12]|files -> rows|exit" | awk '/^run/{w=$6} /files -> rows/{f=$(NF-1)} /exit/{printf "   fresh process: wall %s ms, files->rows %s ms, exit %s ms\n", w, f, $(NF-1)}' | tail -2 >> $OUT
  PFA_REPS=7 python3 $R/tools/e2e_trace.py 16384 2>&1 | tail -1 | sed 's/^/   in process:   /' >> $OUT
done
cat $OUT
