# round 4: the batched profile's push / walk balance again, now that the walk phases run beside the sweeps
set -o pipefail
mkdir -p gpurun_out
run() {
  tag=$1; shift
  timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-extras "$@" > gpurun_out/r04t_$tag.json 2> gpurun_out/r04t_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r04t_$tag.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r04t_$tag.json"))
print("$tag", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], "dense/q", d["dense_levels_per_query"], "walks", d["walks_per_query"], "rounds", d["avg_rounds"], "sweep_us", d["roofline"]["avg_launch_us"], flush=True)
PY
}
run base || exit 1
run w025 --tuning c_walk_ns=0.25 || exit 1
run w018 --tuning c_walk_ns=0.18 || exit 1
run w012 --tuning c_walk_ns=0.12 || exit 1
run w050 --tuning c_walk_ns=0.5 || exit 1
run p12 --tuning prior_levels=12 || exit 1
run p20 --tuning prior_levels=20 || exit 1
run h15 --tuning halving_ratio=1.5 || exit 1
run h30 --tuning halving_ratio=3 || exit 1
run w018p12 --tuning c_walk_ns=0.18,prior_levels=12 || exit 1
run w025h15 --tuning c_walk_ns=0.25,halving_ratio=1.5 || exit 1
run q50base --queries-per-step 50 || exit 1
# one query at a time (the drop-in call): sweep shape and balance once more, with the sparse levels now in one launch
srun() {
  tag=$1; shift
  run s_$tag --mode single --queries-per-step 32 "$@"
}
srun base || exit 1
srun gs3 --tuning gs_blocks=3 || exit 1
srun df10 --tuning dense_frac=0.1 || exit 1
srun df20 --tuning dense_frac=0.2 || exit 1
srun df02 --tuning dense_frac=0.02 || exit 1
srun w025 --tuning c_walk_ns=0.25 || exit 1
srun w050 --tuning c_walk_ns=0.5 || exit 1
srun gs3df10 --tuning gs_blocks=3,dense_frac=0.1 || exit 1
