# round 4, last session: does HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory: shorter launches) help the
# latency-bound paths - top-k one at a time / 16 in flight, one whole-graph query at a time - and the headline?
mkdir -p gpurun_out
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/s3_kernarg.log
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python tools/bench_topk.py 22 128 2>/dev/null >> gpurun_out/s3_kernarg.log
  HIP_FORCE_DEV_KERNARG=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-rmat24 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], 'single', d['one_query_at_a_time']['value'], 'topk', d['topk_sample']['value'], d['topk_sample']['one_at_a_time_queries_per_s'], 'q50', d['value_q50'], d['value_q50_stream'])" >> gpurun_out/s3_kernarg.log
done
cat gpurun_out/s3_kernarg.log
