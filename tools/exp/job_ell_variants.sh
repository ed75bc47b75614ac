#!/bin/bash
# the batched sweep's edge kernel alone, panel copy and row-major (tools/exp/sweep_edges_time.py), one line each
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/${1:-panel_variants}.txt
: > $out
run() { env TAG="$*" "$@" timeout -k 10 180 python3 $root/tools/exp/sweep_edges_time.py 22 10 >> $out 2>&1 || echo "failed: $*" >> $out; }
for d in 0 1 2 3 0x7fff00 0x7fff02 0x3ffff00; do run ONLY_PART=1 PPRHIP_PANEL_DBG=$d; done
cat $out
