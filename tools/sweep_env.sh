# developer helper: bench.py under several values of one environment variable (run on the GPU box)
# usage: sweep_env.sh VAR "extra bench args" v1 v2 ...
var=$1; extra=$2; shift; shift
for v in "$@"; do
  echo "== $var=$v $extra"
  env $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 4 --warmup 1 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['queries_per_s_live_sources'], d['avg_rounds'], d['kernel_ms_per_live_query'], d['dense_levels_per_live_query'], d['levels_per_live_query'])"
done
