mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "batch or fora or smoke or cli or full_size" > gpurun_out/r04q_t.log 2>&1; echo rc=$? >> gpurun_out/r04q_t.log; tail -3 gpurun_out/r04q_t.log
: > gpurun_out/r04q_drivers.log
for d in 1 2 4 8 16 4 1; do
  PPRHIP_BATCH_DRIVERS=$d timeout -k 10 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r04q_b.json 2>> gpurun_out/r04q_drivers.log || { echo "drivers $d failed" >> gpurun_out/r04q_drivers.log; continue; }
  python - >> gpurun_out/r04q_drivers.log <<PY
import json
d=json.load(open("gpurun_out/r04q_b.json"))
print("drivers $d", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"], flush=True)
PY
done
grep drivers gpurun_out/r04q_drivers.log
