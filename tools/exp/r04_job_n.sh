mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04n_t.log 2>&1; echo rc=$? >> gpurun_out/r04n_t.log; tail -3 gpurun_out/r04n_t.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-rmat24 --no-pmc > gpurun_out/r04n_bench.json 2> gpurun_out/r04n_bench.err; echo bench rc=$?
python - <<PY
import json
d=json.load(open("gpurun_out/r04n_bench.json"))
print(d["value"], d["value_q50"], d["value_q50_note"])
PY
