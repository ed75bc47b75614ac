set -o pipefail
PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=50 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -x -q -k "not full_size" > gpurun_out/r02v_t1.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02v_t1.log
grep -q " passed" gpurun_out/r02v_t1.log || exit 1
for cfg in "PPRHIP_HOTB_LINES=1024" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=8192" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=16384" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=24576" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=32768" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=65536" "PPRHIP_HOTB_LINES=0 PPRHIP_L2_KEEP_IDS=4000000"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r02v_$tag.json 2> gpurun_out/r02v_$tag.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r02v_$tag.json"))
print("$cfg", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"], flush=True)
PY
done
