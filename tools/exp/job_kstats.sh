#!/bin/bash
# Per-kernel times of the headline workload (one `rocprofv3 --kernel-trace --stats` pass, two timed steps), for A/B runs:
#   gpurun -- 'tools/exp/job_kstats.sh tag'      -> gpurun_out/<tag>_kstats.txt (top kernels)
set -o pipefail
tag=${1:-kstats}
export PPRHIP_LIB_PATH=${PPRHIP_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/personalized-pagerank-algorithms-on-neo4j_amd/libpprhip_hooks.so}  # (the switches these jobs set are test hooks)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- python3 $root/bench.py ${KSTATS_ARGS:---steps 2 --warmup 1} --no-cpu-baseline --no-pmc --no-extras > /tmp/ks_$tag.log 2>&1 || echo "(the profiled program failed - a measurement switch that breaks the results? - its kernel times follow)"
python3 - "$tag" > $out/${tag}_kstats.txt <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ks_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:10]:
    print("%-60s calls %6s avg %9.1f us total %9.1f ms" % (r["Name"].replace("pprhip::", "").replace("void ", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
grep -o '"value": [0-9.]*' /tmp/ks_$tag.log | head -1 >> $out/${tag}_kstats.txt
cat $out/${tag}_kstats.txt
