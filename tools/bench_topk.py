"""Developer benchmark: FORA top-k (k = 32) queries/s, one query at a time vs 16 in flight (config #3 / #4 shapes)."""
import importlib
import os
import sys
import time

import numpy as np
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py runs
import torch  # noqa: F401  (loads the HIP runtime first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
q = int(sys.argv[2]) if len(sys.argv) > 2 else 64
host = pkg.HostCsr.rmat(scale, 16, seed=1)
g = pkg.Graph(host)
srcs = np.random.default_rng(2).integers(0, host.n, size=q).astype(np.int32)
live = int((np.diff(host.out_rp)[srcs] > 0).sum())
g.fora_batch_topk(srcs[:16], 32, 0.5, 0.15, seed=1)  # warm-up (allocates the slots)
t0 = time.perf_counter()
for i, s in enumerate(srcs):
    g.fora_topk(int(s), 0.5, 0.15, 32, seed=5 + i)
t1 = time.perf_counter()
print("single: %.1f queries/s (%.2f ms per live query)" % (q / (t1 - t0), 1e3 * (t1 - t0) / live), flush=True)
for name, tun in (("default", pkg.tuning_default()), ("batch", pkg.tuning_batch())):
    g.set_tuning(tun)
    t0 = time.perf_counter()
    ids, vals, st = g.fora_batch_topk(srcs, 32, 0.5, 0.15, seed=5)
    t1 = time.perf_counter()
    print("batched (%s profile): %.1f queries/s (%.2f ms per live query), rounds %d, dense levels %d, sweeps %d"
          % (name, q / (t1 - t0), 1e3 * (t1 - t0) / live, st.rounds, st.dense_levels, st.class_launches[5]), flush=True)
g.close()

# walk phases of top-k rounds (short launches): rate and fill of the loads, against the long phases of whole-graph queries
pkg.set_kernel_timing(True)
g = pkg.Graph(host)
tot = {"steps": 0, "loads": 0, "lanes": 0, "ms": 0.0, "launches": 0, "walks": 0}
for i, s in enumerate(srcs[:32]):
    _, _, _, _, st = g.fora_topk(int(s), 0.5, 0.15, 32, seed=5 + i)
    tot["steps"] += st.walk_steps
    tot["loads"] += st.walk_loads
    tot["lanes"] += st.walk_load_lanes
    tot["ms"] += st.class_ms[3]
    tot["launches"] += st.class_launches[3]
    tot["walks"] += st.walks
if tot["launches"]:
    print("top-k walk phases: %d launches, %.0f us each, %.1f G steps/s, %.1f lanes per load, %.0f walks and %.0f steps per launch"
          % (tot["launches"], 1e3 * tot["ms"] / tot["launches"], tot["steps"] / (tot["ms"] / 1e3) / 1e9,
             tot["lanes"] / max(1, tot["loads"]), tot["walks"] / tot["launches"], tot["steps"] / tot["launches"]), flush=True)
g.close()
