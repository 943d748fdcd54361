#!/bin/bash
# read() against mmap() for P processes x T threads over 16 384 page-cached files of 510 kB (tools/ubench/read_vs_mmap.cpp):
# P address spaces as P ranks have.  Host-only (no GPU call).  usage: tools/read_vs_mmap_procs.sh OUTFILE
OUT=${1:-gpurun_out/read_vs_mmap_procs.txt}
R0=$(cd "$(dirname "$0")/.." && pwd)
BIN=/tmp/read_vs_mmap_$$
g++ -O2 -std=c++17 -pthread -o $BIN $R0/tools/ubench/read_vs_mmap.cpp || exit 1
D=/dev/shm/pfa_rvm_$$
mkdir -p $D
{
  echo "# CPUs visible: $(nproc); control-group CPU quota (cpu.max): $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  echo "# 16384 files x 510000 bytes in tmpfs; every line: one pass of all files (3 repeats per mode)"
  $BIN $D 16384 510000 16 1 > /dev/null      # writes the files, warms the page cache
  for cfg in "16 1" "8 2" "4 4" "2 8" "1 16" "16 8" "128 1"; do
    set -- $cfg
    th0=$(grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' ')
    if [ "$2" = "1" ]; then $BIN $D 16384 510000 $1; else $BIN $D 16384 510000 $1 $2; fi
    echo "   (threads $1 x processes $2: $th0 -> $(grep -E "nr_throttled" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '))"
  done
} > "$OUT" 2>&1
rm -rf $D $BIN
cat "$OUT"
