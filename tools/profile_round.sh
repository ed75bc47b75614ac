#!/bin/bash
# Collects the rocprofv3 evidence of a round on the GPU box (run from the repo root through gpurun):
#   tools/profile_round.sh <tag> <what>     what = bench | single | topk | apbs
# For each workload: one `--kernel-trace --stats` run and two PMC runs (FETCH_SIZE, WRITE_SIZE; they do not fit one
# pass on gfx950), the program directly after `--`.  Raw output goes to gpurun_out/<tag>_<what>_{stats,fetch,write};
# tools/summarize_profile.py condenses it into profiles/.
set -o pipefail
tag=$1; what=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
export TMPDIR=/tmp
cd /tmp
case $what in
  bench)  prog="$root/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras" ;;
  single) prog="$root/bench.py --mode single --queries-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras" ;;
  topk)   prog="$root/tools/bench_topk.py 22 64" ;;
  # (until round 3 this workload ran with PPRHIP_BATCH_THREADS=0, after round 1's abort under the profiler with 16
  # launching threads; since round 4 it runs as the product does - DESIGN.md 5, "The abort on record")
  apbs)   prog="$root/tools/explore_apbs.py --scale 22 --thr 1e-3 --targets 262144" ;;
  *) echo "unknown workload $what"; exit 2 ;;
esac
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_${what}_stats -- python3 $prog > $out/${tag}_${what}_stats.log 2>&1 || { echo "stats run failed"; tail -5 $out/${tag}_${what}_stats.log; exit 1; }
grep -v "simple_timer\|generateRocpd\|^W2\|^E2" $out/${tag}_${what}_stats.log | tail -4
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_${what}_fetch -- python3 $prog > $out/${tag}_${what}_fetch.log 2>&1 || { echo "FETCH_SIZE run failed"; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_${what}_write -- python3 $prog > $out/${tag}_${what}_write.log 2>&1 || { echo "WRITE_SIZE run failed"; exit 1; }
# what travels back is the condensed form (gpurun merges at most 64 MiB; the per-dispatch CSVs are ~10 MB each)
cd $root && python3 tools/summarize_profile.py $tag $what > /dev/null && mkdir -p $out/profiles_$tag && cp profiles/${tag}_${what}_* $out/profiles_$tag/
rm -rf $out/${tag}_${what}_stats $out/${tag}_${what}_fetch $out/${tag}_${what}_write
echo "profiled $what"
