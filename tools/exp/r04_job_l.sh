mkdir -p gpurun_out
timeout -k 10 600 python bench.py --no-cpu-baseline --no-rmat24 > gpurun_out/r04l_bench.json 2> gpurun_out/r04l_bench.err; echo bench rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r04l_bench.json"))
print(d["value"], d["value_q50"], d["compute_stream_idle_frac"])
for g in d["stream_occupancy"].get("largest_gaps_on_compute_stream", []): print(g)
print(d["all_pair_sample"]["value"], d["all_pair_scaling"])
PY
