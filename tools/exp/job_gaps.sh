#!/bin/bash
# Kernel trace of the headline's trace child and the examples of its largest compute-stream gaps:
#   gpurun -- 'tools/exp/job_gaps.sh tag'  -> gpurun_out/<tag>_gaps.txt
set -o pipefail
tag=${1:-gaps}
export PPRHIP_LIB_PATH=${PPRHIP_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/personalized-pagerank-algorithms-on-neo4j_amd/libpprhip_hooks.so}  # (the switches these jobs set are test hooks)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/gp_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$tag -- python3 $root/bench.py --trace-child --steps 2 --warmup 1 > /tmp/gp_$tag.log 2>&1 || { tail -5 /tmp/gp_$tag.log; exit 1; }
python3 $root/tools/exp/sweep_gaps.py /tmp/gp_$tag ${2:-3} > $out/${tag}_gaps.txt
python3 $root/tools/exp/gap_timeline.py /tmp/gp_$tag >> $out/${tag}_gaps.txt
head -5 $out/${tag}_gaps.txt
