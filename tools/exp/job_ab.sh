#!/bin/bash
# A/B of the headline under environment switches, one bench run (4 timed steps) per configuration:
#   gpurun -- 'tools/exp/job_ab.sh tag "" "PPRHIP_BATCH_THREADS=1" "PPRHIP_BATCH_THREADS=1 PPRHIP_WORKER_WALK_WIDE=1"'
#   -> gpurun_out/<tag>_ab.txt: one line per configuration (value, ms per step, sweeps per query)
set -o pipefail
tag=$1; shift
export PPRHIP_LIB_PATH=${PPRHIP_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/personalized-pagerank-algorithms-on-neo4j_amd/libpprhip_hooks.so}  # (the switches these jobs set are test hooks)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
: > $out/${tag}_ab.txt
for cfg in "$@"; do
  env $cfg timeout -k 10 300 python3 $root/bench.py --steps ${AB_STEPS:-4} --warmup 1 --no-cpu-baseline --no-pmc --no-extras --no-rmat24 > /tmp/ab.log 2>&1
  rc=$?
  python3 - "$cfg" $rc >> $out/${tag}_ab.txt <<'PY'
import json, sys
cfg, rc = sys.argv[1], sys.argv[2]
line = [l for l in open("/tmp/ab.log") if l.startswith("{")]
if not line:
    print("%-60s rc %s no json: %s" % (cfg or "(default)", rc, open("/tmp/ab.log").read()[-300:].replace("\n", " | ")))
else:
    d = json.loads(line[-1])
    r = d.get("roofline", {})
    print("%-60s value %8.2f ms/step %8.2f sweeps %s avg_sweep_us %s" % (cfg or "(default)", d["value"], d["ms_per_step"], r.get("sweeps"), r.get("avg_sweep_us")))
PY
  [ $rc -ne 0 ] && [ $rc -ne 1 ] && break
done
cat $out/${tag}_ab.txt
