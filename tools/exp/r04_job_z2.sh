# round 4, last session: further runtime switches beside HIP_FORCE_DEV_KERNARG=1 (active waits instead of interrupts)
mkdir -p gpurun_out
run() {
  echo "== $*" >> gpurun_out/s3_kernarg2.log
  env "$@" timeout -k 10 200 python tools/bench_topk.py 22 128 2>/dev/null >> gpurun_out/s3_kernarg2.log
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --no-rmat24 --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('headline', d['value'], 'single', d['one_query_at_a_time']['value'], 'topk', d['topk_sample']['value'], d['topk_sample']['one_at_a_time_queries_per_s'], 'q50', d['value_q50'], d['value_q50_stream'], 'ap', d['all_pair_sample']['value'])" >> gpurun_out/s3_kernarg2.log
}
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=2000
run HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
cat gpurun_out/s3_kernarg2.log
