"""Config #4's per-GPU shares on one GPU: calls of 50 / 25 / 13 / 7 live sources at R-MAT 22 under the batch profile and
under pprhip_tuning_batch_for(q) (what bench.py prints as config4_share_rates), plus dense levels and walks per query."""
import importlib
import os
import sys
import time

import numpy as np
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
host = pkg.HostCsr.rmat(scale, 16, seed=1)
live = np.nonzero(np.diff(host.out_rp) > 0)[0].astype(np.int32)
rng = np.random.default_rng(2)
calls = 4
with pkg.Graph(host) as g:
    store = pkg.Results(g, 64)
    conf = pkg.conf_whole_graph(host.n, host.m, 0.15)
    for q in (50, 25, 13, 7, 3):
        srcs = live[rng.integers(0, live.size, size=(calls + 1, q))].astype(np.int32)
        for name, tun in (("batch", pkg.tuning_batch()), ("for_q", pkg.tuning_batch_for(q))):
            g.set_tuning(tun)
            g.fora_batch_single_source(srcs[0], 0.5, 0.15, seed=61, k=32, conf=conf, keep=store)
            t0 = time.perf_counter()
            dl = wk = 0
            for i in range(1, calls + 1):
                st = g.fora_batch_single_source(srcs[i], 0.5, 0.15, seed=61 + i, k=32, conf=conf, keep=store)[-1]
                dl += st.dense_levels
                wk += st.walks
            dt = (time.perf_counter() - t0) / calls
            print("q=%2d %-6s %7.1f queries/s  %7.2f ms per call  dense levels per query %.1f  walks per query %.3g"
                  % (q, name, q / dt, 1e3 * dt, dl / (calls * q), wk / (calls * q)), flush=True)
    store.close()
