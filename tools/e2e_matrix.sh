#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 tools/e2e_trace.py ${M:-4096} 2>&1 | grep -E "^first|PFA_TRACE" | grep -v "TRACE batch" | cut -c1-250; }
run PFA_X=2 PFA_TRACE_LAST=1
M=8192 run PFA_X=2 PFA_TRACE_LAST=1
M=8192 run PFA_INGEST_SLOT_BYTES=33554432 PFA_INGEST_SLOTS=8 PFA_INGEST_INFLIGHT=4
