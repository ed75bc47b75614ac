#!/bin/bash
# Round 3: per-kernel times of the walk phase (k_mc_plan, k_mc_walk) in the three query workloads, one rocprofv3
# --kernel-trace --stats run each.   gpurun -- tools/exp/job_walk_stats.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for what in topk bench single; do
  case $what in
    bench)  prog="$root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras" ;;
    single) prog="$root/bench.py --mode single --queries-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras" ;;
    topk)   prog="$root/tools/bench_topk.py 22 64" ;;
  esac
  rm -rf /tmp/ws_$what
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ws_$what -- python3 $prog > /tmp/ws_$what.log 2>&1 || { echo "$what failed"; tail -5 /tmp/ws_$what.log; exit 1; }
  echo "== $what"
  grep -E "queries/s|\"value\"" /tmp/ws_$what.log | cut -c1-200
  python3 - "$what" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/ws_%s/**/*kernel_stats.csv" % sys.argv[1], recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:9]:
    print("%-60s calls %6s  total %9.3f ms  avg %9.1f us  %s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
done
