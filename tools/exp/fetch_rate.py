"""Delivery rate of whole vectors to pageable host memory against the resident rate (developer tool, round 3).

    python tools/exp/fetch_rate.py [scale] [queries]

Runs the headline batch three ways on the same sources: vectors kept in the device store, vectors fetched into a
pre-touched host array, and both at once.
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
q = int(sys.argv[2]) if len(sys.argv) > 2 else 128
host = bench.load_host(pkg, scale)
g = pkg.Graph(host, device=0)
g.set_tuning(pkg.tuning_batch())
conf = pkg.conf_whole_graph(host.n, host.m, bench.ALPHA)
live = np.flatnonzero(np.diff(host.out_rp) > 0).astype(np.int32)
rng = np.random.default_rng(5)
s = rng.choice(live, size=q).astype(np.int32)
store = pkg.Results(g, q)
dest = np.zeros((q, host.n))
g.fora_batch_single_source(s[:16], bench.EPS, bench.ALPHA, seed=1, k=32, conf=conf, keep=store)
g.fora_batch_single_source(s[:16], bench.EPS, bench.ALPHA, seed=1, k=32, conf=conf, fetch=True, out=dest[:16])
for name, kw in (("resident", dict(keep=store)), ("fetched", dict(fetch=True, out=dest)), ("resident", dict(keep=store)),
                 ("fetched", dict(fetch=True, out=dest))):
    t0 = time.perf_counter()
    st = g.fora_batch_single_source(s, bench.EPS, bench.ALPHA, seed=2, k=32, conf=conf, **kw)[-1]
    dt = time.perf_counter() - t0
    print("%-9s %7.1f queries/s (%.3f ms per query)  kernel classes ms/query: %s" % (
        name, q / dt, 1e3 * dt / q, {pkg.KERNEL_NAMES[c]: round(st.class_ms[c] / q, 3) for c in (2, 3, 5, 6)}), flush=True)
