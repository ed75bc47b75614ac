"""All-Pair-Backward-Search over ALL targets of R-MAT 24 on one GPU (config #5's graph; developer tool, round 3).

    python tools/exp/apbs_rmat24_all.py [targets]
"""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
t0 = time.time()
host = pkg.HostCsr.rmat(24, 16, seed=1)
print("generated in %.1f s" % (time.time() - t0), flush=True)
nt = int(sys.argv[1]) if len(sys.argv) > 1 else host.n
t0 = time.time()
with pkg.Graph(host, device=0) as g:
    print("lifted in %.1f s" % (time.time() - t0), flush=True)
    ix, _ = g.all_pair_backward(0.15, 1e-3, 32, 0, 4096)
    ix.close()
    t0 = time.perf_counter()
    ix, st = g.all_pair_backward(0.15, 1e-3, 32, 0, nt)
    dt = time.perf_counter() - t0
    off, tg, vl = ix.arrays()
    print("targets %d in %.2f s = %.0f targets/s, entries %d, pops %d, edges %d" % (nt, dt, nt / dt, len(tg), st.pops, st.edge_pushes), flush=True)
    ix.close()
