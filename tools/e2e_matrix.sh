#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" PFA_REPS=11 timeout -k 10 200 python3 tools/e2e_trace.py ${M:-8192} 2>&1 | grep -E "^first" | cut -c1-330; }
for T in 10 12 14 16 20 24 28; do run PFA_THREADS=$T; done
run PFA_THREADS=16 PFA_INGEST_SLOTS=16
run PFA_THREADS=24 PFA_INGEST_SLOTS=16
for T in 16 24; do run PFA_THREADS=$T; done
