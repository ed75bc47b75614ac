# round 4, last session: what do the kernel-class timers (two event records per bracket, in every call) cost the
# latency-bound paths?  PPRHIP_KERNEL_TIMER=0 switches them off everywhere.
mkdir -p gpurun_out
for v in 1 0 1 0; do
  echo "== PPRHIP_KERNEL_TIMER=$v" >> gpurun_out/s3_ktimer.log
  PPRHIP_KERNEL_TIMER=$v timeout -k 10 200 python tools/bench_topk.py 22 128 2>/dev/null >> gpurun_out/s3_ktimer.log
  PPRHIP_KERNEL_TIMER=$v timeout -k 10 200 python bench.py --mode single --queries-per-step 32 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('one query at a time', d['value'])" >> gpurun_out/s3_ktimer.log
done
cat gpurun_out/s3_ktimer.log
