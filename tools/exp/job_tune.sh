set -o pipefail
run() {
  tag=$1; shift
  timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pmc --no-extras "$@" > gpurun_out/r02s_$tag.json 2> gpurun_out/r02s_$tag.err || { echo "$tag failed"; tail -3 gpurun_out/r02s_$tag.err; return 1; }
  python - <<PY
import json
d=json.load(open("gpurun_out/r02s_$tag.json"))
print("$tag", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], "dense/q", d["dense_levels_per_query"], "walks", d["walks_per_query"], "rounds", d["avg_rounds"], "sweep_us", d["roofline"]["avg_launch_us"], flush=True)
PY
}
run base || exit 1
run rounds2 --rounds 2 || exit 1
run rounds3 --rounds 3 || exit 1
run gs4 --tuning gs_blocks=4 || exit 1
run gsf02 --tuning gs_frac=0.02 || exit 1
run gsf10 --tuning gs_frac=0.1 || exit 1
run df01 --tuning dense_frac=0.01 || exit 1
run df04 --tuning dense_frac=0.04 || exit 1
run walk2x --tuning c_walk_ns=$(python - <<PY
import importlib
pkg=importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
print(pkg.tuning_batch().c_walk_ns*2)
PY
) || exit 1
