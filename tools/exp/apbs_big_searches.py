"""Round 4: where should All-Pair's largest searches run?  For the targets with the most in-edges of R-MAT 22: the
whole-vector backward search (pprhip_backward_push: sparse levels + pull sweeps over the out-CSR, the whole chip on one
search) against the same target through pprhip_all_pair_backward on a range of one (LDS tier gives up -> dense tier ->
full-size pass), and the dense tier's own phase census for the whole range (PPRHIP_APBS_DEBUG)."""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: F401,E402
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
from bench import load_host  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
host = load_host(pkg, scale)
ind = np.diff(host.in_rp).astype(np.int64)
order = np.argsort(-ind, kind="stable")
A, THR = 0.15, 1e-3
with pkg.Graph(host) as g:
    ix, _ = g.all_pair_backward(A, THR, 32, 0, 4096)
    ix.close()
    g.backward_push(int(order[0]), A, THR)
    print("rank target in_deg | whole-vector ms (levels, dense levels, pops, edges) | all_pair(range of 1) ms (edges)")
    for rank in (0, 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
        t = int(order[rank])
        t0 = time.perf_counter()
        p, r, st = g.backward_push(t, A, THR)
        a = time.perf_counter() - t0
        t0 = time.perf_counter()
        ix, st2 = g.all_pair_backward(A, THR, 32, t, t + 1)
        b = time.perf_counter() - t0
        ix.close()
        print("%6d %8d %7d | %8.2f ms dev %.2f (%d lv, %d dense, %d pops + %d dense nodes, %d edges) | %8.2f ms dev %.2f (%d edges, xl %d)"
              % (rank, t, ind[t], 1e3 * a, st.total_ms, st.levels, st.dense_levels, st.pops, st.dense_nodes, st.edge_pushes,
                 1e3 * b, st2.total_ms, st2.edge_pushes, st2.xl_targets), flush=True)
