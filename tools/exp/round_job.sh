#!/bin/bash
# One parametrised GPU job instead of a script per run (rounds 2-4 left thirty of those; git history has them).
# Run from the repo root through gpurun; every step writes under gpurun_out/<tag>_* and prints a short tail.
#   tools/exp/round_job.sh <tag> <step> [<step> ...]       steps are run in order, the job stops at the first failure
# steps:
#   tests[:<pytest -k expression>]   the GPU tests (all, or a selection)
#   bench[:<extra bench.py flags>]   bench.py (default flags: the driver's), e.g. bench:--no-cpu-baseline_--no-pmc
#   sweep:<VAR>=<v1>,<v2>,...        the short headline bench once per value of an environment switch
#   allpair22 | allpair24            All-Pair on all targets of R-MAT 22 / 24 with PPRHIP_APBS_DEBUG phase lines
#   topk                             tools/bench_topk.py 22 256
#   profile:<bench|single|topk|apbs> tools/profile_round.sh (rocprofv3 stats + FETCH_SIZE + WRITE_SIZE passes)
#   kstats                           tools/exp/job_kstats.sh (per-kernel times of the headline workload)
#   gloo2                            bench.py --gpus 2 over gloo on the one GPU (rehearsal of the N > 1 launch path)
# Environment switches of the library (PPRHIP_*) set for the job apply to every step.
set -o pipefail
tag=$1; shift
mkdir -p gpurun_out
short="--steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-extras"
headline() {  # prints value, ms per query, kernel classes, sweep time of a bench.py line
  python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(d["value"], d["ms_per_query"], d.get("kernel_ms_per_query"), r.get("avg_sweep_us", r.get("avg_launch_us")), flush=True)
PY
}
for step in "$@"; do
  name=${step%%:*}; arg=""; [[ $step == *:* ]] && arg=${step#*:}
  case $name in
    tests)
      timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 ${arg:+-k "$arg"} > gpurun_out/${tag}_tests.log 2>&1
      rc=$?; echo rc=$rc >> gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log; [ $rc -eq 0 ] || exit 1 ;;
    bench)
      timeout -k 10 1000 python bench.py ${arg//_/ } > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || { tail -5 gpurun_out/${tag}_bench.err; exit 1; }
      headline gpurun_out/${tag}_bench.json ;;
    sweep)
      var=${arg%%=*}; vals=${arg#*=}
      for v in ${vals//,/ }; do
        env $var=$v timeout -k 10 300 python bench.py $short > gpurun_out/${tag}_sweep.json 2>> gpurun_out/${tag}_sweep.err || { echo "$var=$v failed"; exit 1; }
        echo -n "$var=$v " | tee -a gpurun_out/${tag}_sweep.txt; headline gpurun_out/${tag}_sweep.json | tee -a gpurun_out/${tag}_sweep.txt
      done ;;
    allpair22)
      PPRHIP_APBS_DEBUG=1 timeout -k 10 300 python tools/bench_allpair.py --targets-per-rank 4194304 2>&1 | grep -E "apbs|index\]|metric|tier|searches" | cut -c1-300 > gpurun_out/${tag}_ap22.log
      tail -6 gpurun_out/${tag}_ap22.log ;;
    allpair24)
      PPRHIP_APBS_DEBUG=1 timeout -k 10 500 python tools/exp/apbs_rmat24_all.py 2>&1 | grep -E "apbs|index\]|targets|tier|searches" | cut -c1-300 > gpurun_out/${tag}_ap24.log
      tail -6 gpurun_out/${tag}_ap24.log ;;
    topk)
      timeout -k 10 300 python tools/bench_topk.py 22 256 > gpurun_out/${tag}_topk.log 2>&1 || { tail -3 gpurun_out/${tag}_topk.log; exit 1; }
      tail -4 gpurun_out/${tag}_topk.log ;;
    profile)
      bash tools/profile_round.sh $tag $arg > gpurun_out/${tag}_prof_$arg.log 2>&1 || { tail -5 gpurun_out/${tag}_prof_$arg.log; exit 1; }
      tail -2 gpurun_out/${tag}_prof_$arg.log ;;
    kstats)
      bash tools/exp/job_kstats.sh $tag | head -8 ;;
    gloo2)
      PPRHIP_BENCH_WATCHDOG_S=200 timeout -k 10 800 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo > gpurun_out/${tag}_gloo2.json 2> gpurun_out/${tag}_gloo2.err
      echo rc=$?; tail -c 800 gpurun_out/${tag}_gloo2.json ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
