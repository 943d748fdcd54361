#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" PFA_REPS=9 timeout -k 10 200 python3 tools/e2e_trace.py ${M:-16384} 2>&1 | grep -E "^first" | cut -c1-330; }
for i in 1 2; do for T in 16 20 24; do run PFA_THREADS=$T; done; done
run PFA_THREADS=16 PFA_TRACE_LAST=1
