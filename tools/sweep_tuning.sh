# developer helper: bench.py under several cost-model settings (run on the GPU box)
for t in "$@"; do
  echo "== $t"
  timeout -k 10 300 python bench.py --no-cpu-baseline --queries-per-step 128 --steps 4 --warmup 1 --tuning "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['queries_per_s_live_sources'], d['avg_rounds'], d['kernel_ms_per_live_query'], d['dense_levels_per_live_query'], d['roofline']['kernel'])"
done
