"""What a process pays once: graph lift, the first call of each entry point (workspaces, code objects) against the second.
python tools/exp/first_use.py [scale]"""
import importlib
import os
import sys
import time

import numpy as np

t_start = time.time()
import torch  # noqa: F401,E402  (loads the HIP runtime first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
print("imports: %.2f s" % (time.time() - t_start), flush=True)
t = time.time()
host = pkg.HostCsr.rmat(scale, 16, seed=1)
print("generate + two CSRs: %.2f s" % (time.time() - t), flush=True)
live = np.nonzero(np.diff(host.out_rp) > 0)[0]
rng = np.random.default_rng(2)
srcs = live[rng.integers(0, live.size, size=(4, 50))].astype(np.int32)


def timed(label, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = fn()
    torch.cuda.synchronize()
    print("%-44s %8.1f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
    return r


g = timed("pprhip_graph_create (first of the process)", lambda: pkg.Graph(host, device=0))
conf = pkg.conf_whole_graph(host.n, host.m, 0.15)
for i in range(2):
    timed("fora_single_source, call %d" % i, lambda: g.fora_single_source(int(srcs[0, i]), 0.5, 0.15, seed=3, conf=conf, fetch=False))
g.set_tuning(pkg.tuning_batch())
for i in range(3):
    timed("fora_batch_single_source(50), call %d" % i,
          lambda: g.fora_batch_single_source(srcs[i], 0.5, 0.15, seed=3 + i, k=32, conf=conf))
g.set_tuning(pkg.tuning_default())
for i in range(2):
    timed("fora_topk k=32, call %d" % i, lambda: g.fora_topk(int(srcs[1, i]), 0.5, 0.15, 32, seed=5))
for i in range(2):
    timed("fora_batch_topk(50), call %d" % i, lambda: g.fora_batch_topk(srcs[2], 32, 0.5, 0.15, seed=7))
for i in range(3):
    def ap():
        ix, _ = g.all_pair_backward(0.15, 1e-3, 32, 0, 1 << 16)
        ix.close()
    timed("all_pair_backward(2^16 targets), call %d" % i, ap)
timed("graph close", g.close)
