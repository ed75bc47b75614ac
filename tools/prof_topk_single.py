"""Developer helper: a run of single top-k queries for rocprofv3 (kernel mix of pprhip_fora_topk)."""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
host = pkg.HostCsr.rmat(22, 16, seed=1)
g = pkg.Graph(host)
srcs = [int(s) for s in np.random.default_rng(2).integers(0, host.n, size=64)]
live = int((np.diff(host.out_rp)[srcs] > 0).sum())
t0 = time.perf_counter()
rounds = 0
for i, s in enumerate(srcs):
    _, _, _, _, st = g.fora_topk(s, 0.5, 0.15, 32, seed=5 + i)
    rounds += st.rounds
dt = time.perf_counter() - t0
print("single top-k: %.2f ms per live query, %.1f rounds per live query" % (1e3 * dt / live, rounds / live))
g.close()
