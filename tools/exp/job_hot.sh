set -o pipefail
for cfg in "PPRHIP_HOT_IDS=16384" "PPRHIP_HOT_IDS=8192 PPRHIP_EDGE_WGS=2" "PPRHIP_HOT_IDS=8192 PPRHIP_EDGE_WGS=1" "PPRHIP_HOT_IDS=4096 PPRHIP_EDGE_WGS=2" "PPRHIP_HOT_IDS=0 PPRHIP_EDGE_WGS=2" "PPRHIP_HOT_IDS=0 PPRHIP_EDGE_WGS=1"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 300 python bench.py --mode single --queries-per-step 16 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r02p_$tag.json 2> gpurun_out/r02p_$tag.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r02p_$tag.json"))
print("$cfg", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"])
PY
done
