#!/bin/bash
# Headline under cost-model overrides (bench.py --tuning), one run of 4 timed steps per setting:
#   gpurun -- 'tools/exp/job_tune2.sh tag "" "c_walk_ns=0.5" "dense_frac=0.01,c_walk_ns=0.5"'  -> gpurun_out/<tag>_tune.txt
set -o pipefail
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
: > $out/${tag}_tune.txt
for t in "$@"; do
  timeout -k 10 300 python3 $root/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-extras --no-rmat24 ${t:+--tuning $t} > /tmp/tune.log 2>&1
  python3 - "$t" >> $out/${tag}_tune.txt <<'PY'
import json, sys
line = [l for l in open("/tmp/tune.log") if l.startswith("{")]
if not line:
    print("%-44s failed: %s" % (sys.argv[1], open("/tmp/tune.log").read()[-200:].replace("\n", " | ")))
else:
    d = json.loads(line[-1])
    print("%-44s value %7.2f dense/q %5.1f levels/q %5.1f walks/q %8d rounds %.2f sweeps %5d sweep_us %7.1f kernel_ms %s" % (
        sys.argv[1] or "(default)", d["value"], d["dense_levels_per_query"], d["levels_per_query"], d["walks_per_query"], d["avg_rounds"],
        d["roofline"]["sweeps"], d["roofline"]["avg_sweep_us"], d["kernel_ms_per_query"]))
PY
done
cat $out/${tag}_tune.txt
