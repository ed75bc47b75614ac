set -o pipefail
PPRHIP_SLICE_IDS=37 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference.py tests/test_gpu_cli.py -m gpu -x -q > gpurun_out/r02o_t1.log 2>&1; echo "forced-slices rc=$?"; tail -3 gpurun_out/r02o_t1.log
grep -q " passed" gpurun_out/r02o_t1.log || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "not full_size" > gpurun_out/r02o_t2.log 2>&1; echo "scale rc=$?"; tail -3 gpurun_out/r02o_t2.log
grep -q " passed" gpurun_out/r02o_t2.log || exit 1
for cfg in "PPRHIP_SLICED=0" "PPRHIP_SLICE_IDS=262144" "PPRHIP_SLICE_IDS=393216" "PPRHIP_SLICE_IDS=524288"; do
  env $cfg timeout -k 10 300 python bench.py --mode single --queries-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r02o_single_$cfg.json 2> gpurun_out/r02o_single_$cfg.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r02o_single_$cfg.json"))
print("$cfg", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"], d.get("graph_lift_s"))
PY
done
