#!/bin/bash
# Round 3: the one-query-at-a-time path (pprhip_fora_single_source) under other sweep shapes and push / walk balances:
# Gauss-Seidel blocks, the frontier share from which a level runs dense, the modelled cost of a walk.
# One line per setting: queries/s, ms per query, dense levels per query.   gpurun -- tools/exp/job_tune_single.sh
cd "${GRAFT_REPO_ROOT:-.}"
run() {
  python bench.py --mode single --queries-per-step 24 --steps 2 --warmup 1 --no-extras --no-pmc --no-cpu-baseline --tuning "$1" 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-60s %7.2f q/s  %6.3f ms  kernels %s' % (sys.argv[1] or 'default', d['value'], d['ms_per_query'], d['kernel_ms_per_query']))" "$1"
}
run ""
for t in gs_blocks=3 gs_blocks=4 dense_frac=0.03 dense_frac=0.02 gs_blocks=3,dense_frac=0.03 c_walk_ns=0.25 c_walk_ns=0.30 c_walk_ns=0.45 c_walk_ns=0.55 \
         gs_blocks=3,c_walk_ns=0.45 gs_frac=0.05 gs_blocks=3,gs_frac=0.05,dense_frac=0.03; do
  run "$t"
done
