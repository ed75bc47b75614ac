"""A/B of the top-k push-ahead on live sources, with and without the delivery pipe's copy stream on the handle
(developer tool, round 3).   PPRHIP_TOPK_AHEAD=0|1 python tools/exp/topk_ahead_ab.py [pipe]"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
host = bench.load_host(pkg, 22)
g = pkg.Graph(host, device=0)
live = np.flatnonzero(np.diff(host.out_rp) > 0).astype(np.int32)
s = np.random.default_rng(2).choice(live, size=48).astype(np.int32)
if len(sys.argv) > 1 and sys.argv[1] == "pipe_after":
    g.fora_topk(int(s[0]), 0.5, 0.15, 32, seed=1)  # the second stream exists before the pipe's copy stream
if len(sys.argv) > 1 and sys.argv[1] in ("pipe", "pipe_after"):
    g.set_tuning(pkg.tuning_batch())
    conf = pkg.conf_whole_graph(host.n, host.m, bench.ALPHA)
    dest = np.zeros((16, host.n))
    g.fora_batch_single_source(s[:16], bench.EPS, bench.ALPHA, seed=1, k=32, conf=conf, fetch=True, out=dest)
    g.set_tuning(pkg.tuning_default())
elif len(sys.argv) > 1 and sys.argv[1] == "slots":
    g.fora_batch_topk(s[:16], 32, 0.5, 0.15, seed=1)
g.fora_topk(int(s[0]), 0.5, 0.15, 32, seed=1)
for rep in range(2):
    t0 = time.perf_counter()
    rounds = 0
    for j, v in enumerate(s[:32]):
        rounds += g.fora_topk(int(v), 0.5, 0.15, 32, seed=7 + j)[-1].rounds
    dt = time.perf_counter() - t0
    print("ahead=%s %s: %.1f queries/s (%.2f ms per query, %.2f rounds)" % (os.environ.get("PPRHIP_TOPK_AHEAD", "1"), " ".join(sys.argv[1:]) or "plain", 32 / dt, 1e3 * dt / 32, rounds / 32), flush=True)
g.close()
