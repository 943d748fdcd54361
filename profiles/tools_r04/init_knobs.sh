#!/bin/bash
# hipInit under environment knobs (fresh process each, 4 repeats): is any of the runtime's start-up optional for a one-file run?
R=${GRAFT_REPO_ROOT:-$(pwd)}
P=$R/tools/ubench/init_probe
run() {
  local label=$1; shift
  for i in 1 2 3 4; do
    echo "$label :: $(env "$@" $P | grep -E "^hipInit|^total" | awk '{printf "%s %s ms  ", $1, $(NF-1)}' | sed 's/(gfx950.*//')"
  done
}
ls /sys/class/kfd/kfd/topology/nodes/ | tr '\n' ' '; echo
run default A=1
run sdma_off HSA_ENABLE_SDMA=0
run rocr_visible_0 ROCR_VISIBLE_DEVICES=0
run hip_visible_0 HIP_VISIBLE_DEVICES=0
run no_interrupt HSA_ENABLE_INTERRUPT=0
run bind_now LD_BIND_NOW=1
run all3 HSA_ENABLE_SDMA=0 ROCR_VISIBLE_DEVICES=0 HSA_ENABLE_INTERRUPT=0
LD_DEBUG=statistics $P 2>&1 | grep -E "total startup time|relocation|load" | head -8
ldd $P | wc -l
