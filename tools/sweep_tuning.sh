for t in "c_dense_edge_ns=0.012" "c_dense_edge_ns=0.004,c_dense_node_ns=0.006" "c_dense_edge_ns=0.002,c_dense_node_ns=0.003" "c_dense_edge_ns=0.001,c_dense_node_ns=0.002" "c_dense_edge_ns=0.002,c_dense_node_ns=0.003,c_level_ns=4000" "c_dense_edge_ns=0.002,c_dense_node_ns=0.003,c_walk_ns=0.5"; do
  echo "== $t"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --tuning "$t" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['queries_per_s_live_sources'], d['avg_rounds'], d['kernel_ms_per_live_query'], d['dense_levels_per_live_query'])"
done
