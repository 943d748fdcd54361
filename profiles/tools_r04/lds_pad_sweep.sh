#!/bin/bash
# wavefronts per CU against throughput with everything else equal: unused LDS added to every launch (PFA_LDS_PAD).
# usage: tools/lds_pad_sweep.sh OUTFILE
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$1
ROUNDS=1 $R/tools/ab_env.sh $O.n1000 "n1000" "" "PFA_LDS_PAD=256" "PFA_LDS_PAD=768" "PFA_LDS_PAD=1280" "PFA_LDS_PAD=2560" "PFA_LDS_PAD=3840" "PFA_LDS_PAD=5120" "PFA_LDS_PAD=8192" "PFA_LDS_PAD=12288" > /dev/null
ROUNDS=1 $R/tools/ab_env.sh $O.n1000cds "n1000cds" "" "PFA_LDS_PAD=256" "PFA_LDS_PAD=1280" "PFA_LDS_PAD=2560" "PFA_LDS_PAD=5120" "PFA_LDS_PAD=10240" > /dev/null
ROUNDS=1 $R/tools/ab_env.sh $O.headline "headline" "" "PFA_LDS_PAD=1280" "PFA_LDS_PAD=6144" "PFA_LDS_PAD=14336" "PFA_LDS_PAD=28672" > /dev/null
ROUNDS=1 $R/tools/ab_env.sh $O.n200cds "n200cds c4" "" "PFA_LDS_PAD=1280" "PFA_LDS_PAD=6144" "PFA_LDS_PAD=14336" > /dev/null
cat $O.n1000 $O.n1000cds $O.headline $O.n200cds > $O
cat $O
