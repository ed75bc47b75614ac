"""Experiment: time of the sparse push class with / without the LDS combining table (PPRHIP_COMB_MIN_EDGES),
levels forced sparse up to larger sizes by raising dense_frac.  Run once per setting (the knob is read once)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
A = 0.15
host = pkg.HostCsr.rmat(22, 16, seed=1)
od = np.diff(host.out_rp)
rng = np.random.default_rng(3)
srcs = [int(s) for s in rng.integers(0, host.n, 200) if od[s] > 0][:12]
conf = pkg.conf_whole_graph(host.n, host.m, A)
rmax0, _ = pkg.fora_whole_params(conf, 0.5)
with pkg.Graph(host) as g:
    for frac in (None, 0.5):
        t = pkg.tuning_default()
        if frac:
            t.dense_frac = frac
        g.set_tuning(t)
        for rep in range(2):
            ms = 0.0
            ln = 0
            wall = time.perf_counter()
            for s in srcs:
                p, r, rsum, st = g.forward_push(s, A, rmax0)
                ms += st.class_ms[2]
                ln += st.class_launches[2]
            wall = time.perf_counter() - wall
        print("comb_min=%s dense_frac=%s sparse_class_ms/query=%.3f launches=%d wall_ms/query=%.2f" % (
            os.environ.get("PPRHIP_COMB_MIN_EDGES", "default"), frac, ms / len(srcs), ln, wall * 1e3 / len(srcs)), flush=True)
