"""Workload for profiling: FORA top-32 on live sources, 16 in flight (developer tool, round 3)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
q = int(sys.argv[1]) if len(sys.argv) > 1 else 256
host = bench.load_host(pkg, 22)
g = pkg.Graph(host, device=0)
live = np.flatnonzero(np.diff(host.out_rp) > 0).astype(np.int32)
s = np.random.default_rng(5).choice(live, size=q).astype(np.int32)
g.fora_batch_topk(s[:16], 32, 0.5, 0.15, seed=1)
t0 = time.perf_counter()
ids, vals, st = g.fora_batch_topk(s, 32, 0.5, 0.15, seed=5)
dt = time.perf_counter() - t0
print("batched top-k: %.1f queries/s on %d live sources, %.2f rounds per query" % (q / dt, q, st.rounds / q), flush=True)
g.close()
