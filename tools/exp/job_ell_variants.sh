#!/bin/bash
# edge-kernel variants of the source-partitioned sweep, alone (tools/exp/sweep_edges_time.py), one line each
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/${1:-ell_variants}.txt
: > $out
run() { env TAG="$*" "$@" timeout -k 10 120 python3 $root/tools/exp/sweep_edges_time.py 22 10 >> $out 2>&1 || echo "failed: $*" >> $out; }
for d in 0 1 2 3 4 5 7; do run ONLY_PART=1 PPRHIP_ELL_DBG=$d; done
run ONLY_PART=1 PPRHIP_ELL_DBG=1 PPRHIP_ELL_MASK=0x7fff
run ONLY_PART=1 PPRHIP_ELL_DBG=5 PPRHIP_ELL_MASK=0x7fff
cat $out
