set -o pipefail
for cfg in "PPRHIP_HOTB_LINES=0 PPRHIP_EDGEB_WGS=1 PPRHIP_EDGEB_THREADS=1024" "PPRHIP_HOTB_LINES=0 PPRHIP_EDGEB_WGS=3 PPRHIP_EDGEB_THREADS=512" "PPRHIP_HOTB_LINES=0 PPRHIP_EDGEB_WGS=2 PPRHIP_EDGEB_THREADS=512" "PPRHIP_HOTB_LINES=0 PPRHIP_EDGEB_WGS=1 PPRHIP_EDGEB_THREADS=512" "PPRHIP_HOTB_LINES=0 PPRHIP_EDGEB_WGS=6 PPRHIP_EDGEB_THREADS=256"; do
  tag=$(echo $cfg | tr ' =' '__')
  env $cfg timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-extras > gpurun_out/r02u_$tag.json 2> gpurun_out/r02u_$tag.err || exit 1
  python - <<PY
import json
d=json.load(open("gpurun_out/r02u_$tag.json"))
print("$cfg", d["value"], d["ms_per_query"], d["kernel_ms_per_query"], d["roofline"]["avg_launch_us"], flush=True)
PY
done
