#!/bin/bash
# one-query-at-a-time FORA (bench.py --mode single) under cost-model overrides: one line per setting
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/${1:-tune_single}.txt; shift
: > $out
for t in "$@"; do
  timeout -k 10 200 python3 $root/bench.py --mode single --steps 2 --warmup 1 --queries-per-step 16 --no-cpu-baseline --no-pmc --no-extras --no-rmat24 ${t:+--tuning $t} > /tmp/ts.log 2>&1
  python3 - "$t" >> $out <<'PY'
import json, sys
l = [x for x in open("/tmp/ts.log") if x.startswith("{")]
if not l:
    print("%-50s failed" % sys.argv[1])
else:
    d = json.loads(l[-1])
    print("%-50s %7.2f queries/s  dense levels/query %s  walks/query %.3g" % (sys.argv[1] or "(default)", d["value"], d.get("dense_levels_per_query"), d.get("walks_per_query", 0)))
PY
done
cat $out
