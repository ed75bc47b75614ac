# round 4: sparse levels in one launch (k_sparse_levels) against launches per step, by the largest grid the launch may have
mkdir -p gpurun_out
: > gpurun_out/r04d_levels.log
for cfg in "0 128" "1 1" "1 4" "1 16" "1 64" "1 128"; do
  set -- $cfg
  echo "== PPRHIP_SPARSE_LEVELS=$1 GMAX=$2" >> gpurun_out/r04d_levels.log
  PPRHIP_SPARSE_LEVELS=$1 PPRHIP_SPARSE_LEVELS_GMAX=$2 timeout -k 10 200 python tools/bench_topk.py 22 256 >> gpurun_out/r04d_levels.log 2>&1 || exit 1
  PPRHIP_SPARSE_LEVELS=$1 PPRHIP_SPARSE_LEVELS_GMAX=$2 timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r04d_b.json 2>> gpurun_out/r04d_levels.log || exit 1
  python - >> gpurun_out/r04d_levels.log <<PY
import json
d=json.load(open("gpurun_out/r04d_b.json"))
print("headline", d["value"], d["kernel_ms_per_query"], flush=True)
PY
  PPRHIP_SPARSE_LEVELS=$1 PPRHIP_SPARSE_LEVELS_GMAX=$2 timeout -k 10 200 python bench.py --mode single --queries-per-step 32 --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r04d_b.json 2>> gpurun_out/r04d_levels.log || exit 1
  python - >> gpurun_out/r04d_levels.log <<PY
import json
d=json.load(open("gpurun_out/r04d_b.json"))
print("single", d["value"], d["kernel_ms_per_query"], flush=True)
PY
done
timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04d_ap22.log 2>&1; echo rc=$? >> gpurun_out/r04d_ap22.log
PPRHIP_APBS_DEBUG=1 timeout -k 10 200 python tools/bench_allpair.py --targets-per-rank 4194304 > gpurun_out/r04d_apdbg.log 2>&1; echo rc=$? >> gpurun_out/r04d_apdbg.log
timeout -k 10 400 python tools/exp/apbs_rmat24_all.py > gpurun_out/r04d_ap24.log 2>&1; echo rc=$? >> gpurun_out/r04d_ap24.log
bash tools/exp/r04_job_tune.sh > gpurun_out/r04d_tune.log 2>&1; echo rc=$? >> gpurun_out/r04d_tune.log
