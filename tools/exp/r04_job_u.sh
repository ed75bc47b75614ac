mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "batch or fora or smoke or full_size or sweep or gauss" > gpurun_out/r04u_t.log 2>&1; echo rc=$? >> gpurun_out/r04u_t.log; tail -3 gpurun_out/r04u_t.log
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r04u_b.json 2> gpurun_out/r04u_b.err
python - <<PY
import json
d=json.load(open("gpurun_out/r04u_b.json"))
print("run", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"], flush=True)
PY
done
bash tools/profile_round.sh r04 bench > gpurun_out/r04u_prof.log 2>&1; head -12 profiles/r04_bench_summary.md | cut -c1-150
