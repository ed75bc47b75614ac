"""Phase times of the graph lift (PPRHIP_LIFT_DEBUG) for an R-MAT graph: python tools/exp/lift_time.py 22 [threads...]"""
import importlib
import os
import sys
import time

import torch  # noqa: F401  (loads the HIP runtime first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PPRHIP_LIFT_DEBUG"] = "1"
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
t = time.time()
host = pkg.HostCsr.rmat(scale, 16, seed=1)
print("generate + two CSRs: %.2f s" % (time.time() - t), flush=True)
for rep in range(3):
    t = time.time()
    g = pkg.Graph(host, device=0)
    print("pprhip_graph_create (call %d): %.3f s" % (rep, time.time() - t), flush=True)
    g.close()
