"""Experiment (round 4, second session): does a second, independent group of 16 batched queries on the same GPU fill
the compute stream's idle time (compute_stream_idle_frac 0.12-0.18) and hide the apply kernel's latency-bound phases?
Two graph handles (two CSR replicas) on device 0, one host thread each, against one handle with the same total."""
import importlib
import os
import sys
import threading
import time

import numpy as np
import torch  # noqa: F401  (loads the HIP runtime first)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_groups = int(sys.argv[3]) if len(sys.argv) > 3 else 2
Q = 128
EPS, ALPHA, TOPK = 0.5, 0.15, 32
host = pkg.HostCsr.rmat(scale, 16, seed=1)
live = np.nonzero(np.diff(host.out_rp) > 0)[0]
rng = np.random.default_rng(2)
srcs = live[rng.integers(0, live.size, size=(steps + 1, Q))].astype(np.int32)
conf = pkg.conf_whole_graph(host.n, host.m, ALPHA)

graphs, stores = [], []
for i in range(n_groups):
    g = pkg.Graph(host, device=0)
    g.set_tuning(pkg.tuning_batch())
    graphs.append(g)
    stores.append(pkg.Results(g, Q))
    g.fora_batch_single_source(srcs[0], EPS, ALPHA, seed=3, k=TOPK, conf=conf, keep=stores[i])  # warm-up
    print("handle %d ready" % i, flush=True)


def run(gi, blocks, out):
    g = graphs[gi]
    sums = []
    for b, s in enumerate(blocks):
        _, ids, vals, nsel, _, st = g.fora_batch_single_source(s, EPS, ALPHA, seed=5 + b, k=TOPK, conf=conf, keep=stores[gi])
        sums.append((st.dense_levels, st.class_ms[5], st.class_launches[5]))
    out[gi] = sums


# (a) one handle, Q queries per call
torch.cuda.synchronize()
t0 = time.perf_counter()
out = {}
run(0, [srcs[1 + i] for i in range(steps)], out)
torch.cuda.synchronize()
t1 = time.perf_counter()
nq = steps * Q
print("one handle : %.1f queries/s (%d queries, %.1f ms/step), dense levels %d, sweep class %.1f ms / %d launches"
      % (nq / (t1 - t0), nq, 1e3 * (t1 - t0) / steps, sum(x[0] for x in out[0]), sum(x[1] for x in out[0]),
         sum(x[2] for x in out[0])), flush=True)

# (b) n_groups handles beside each other, the same queries dealt out between them (Q / n_groups per call)
per = Q // n_groups
out = {}
ths = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for gi in range(n_groups):
    blocks = [srcs[1 + i][gi * per:(gi + 1) * per] for i in range(steps)]
    th = threading.Thread(target=run, args=(gi, blocks, out))
    th.start()
    ths.append(th)
for th in ths:
    th.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("%d handles, %d queries per call each: %.1f queries/s; sweep class ms per handle %s" %
      (n_groups, per, nq / (t1 - t0), [round(sum(x[1] for x in out[g])) for g in range(n_groups)]), flush=True)

# (c) n_groups handles, Q queries per call each (longer calls: the tail of a call weighs less)
out = {}
ths = []
torch.cuda.synchronize()
t0 = time.perf_counter()
for gi in range(n_groups):
    blocks = [np.roll(srcs[1 + i], 7 * gi) for i in range(steps)]
    th = threading.Thread(target=run, args=(gi, blocks, out))
    th.start()
    ths.append(th)
for th in ths:
    th.join()
torch.cuda.synchronize()
t1 = time.perf_counter()
print("%d handles, %d queries per call each: %.1f queries/s; sweep class ms per handle %s, launches %s" %
      (n_groups, Q, n_groups * nq / (t1 - t0), [round(sum(x[1] for x in out[g])) for g in range(n_groups)],
       [sum(x[2] for x in out[g]) for g in range(n_groups)]), flush=True)
for g in graphs:
    g.close()
