#!/bin/bash
# Duration distribution of one kernel's launches in the headline workload (rocprofv3 --kernel-trace, two timed steps):
#   gpurun -- 'tools/exp/job_ktrace_hist.sh tag k_dense_edges_ell k_dense_apply_batch'
#   -> gpurun_out/<tag>_khist.txt: per kernel name, launches grouped by grid size with min / median / mean / max us
set -o pipefail
tag=${1:-khist}; shift
export PPRHIP_LIB_PATH=${PPRHIP_LIB_PATH:-${GRAFT_REPO_ROOT:-$(pwd)}/personalized-pagerank-algorithms-on-neo4j_amd/libpprhip_hooks.so}  # (the switches these jobs set are test hooks)
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kh_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/kh_$tag -- python3 $root/bench.py ${KSTATS_ARGS:---steps 2 --warmup 1} --no-cpu-baseline --no-pmc --no-extras > /tmp/kh_$tag.log 2>&1 || echo "(the profiled program failed)"
python3 - "$tag" "$@" > $out/${tag}_khist.txt <<'PY'
import csv, glob, sys, collections, statistics
f = glob.glob("/tmp/kh_%s/**/*kernel_trace.csv" % sys.argv[1], recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"].replace("pprhip::", "").replace("void ", "")
    for want in sys.argv[2:]:
        if name.startswith(want):
            acc[(name.split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (name, grid), v in sorted(acc.items()):
    v.sort()
    dec = " ".join("%7.1f" % v[min(len(v) - 1, int(len(v) * q / 10))] for q in range(11))
    print("%-40s grid %8s launches %5d mean %8.1f us; deciles: %s" % (name[:40], grid, len(v), sum(v) / len(v), dec))
PY
grep -o '"value": [0-9.]*' /tmp/kh_$tag.log | head -1 >> $out/${tag}_khist.txt
cat $out/${tag}_khist.txt
