"""Times All-Pair-Backward-Search on an R-MAT graph (developer tool)."""
import argparse, importlib, os, sys, time
import numpy as np
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py runs
import torch  # noqa: F401,E402  (loads the HIP runtime first)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
pkg.set_kernel_timing(True)  # class times wanted here
ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=18)
ap.add_argument("--thr", type=float, default=1e-3)
ap.add_argument("--k", type=int, default=32)
ap.add_argument("--targets", type=int, default=65536)
ap.add_argument("--cpu", type=int, default=0, help="time the CPU oracle on this many targets too")
a = ap.parse_args()
host = pkg.HostCsr.rmat(a.scale, 16, seed=1)
g = pkg.Graph(host)
nt = min(a.targets, host.n)
for rep in range(2):
    t0 = time.time()
    ix, st = g.all_pair_backward(0.15, a.thr, a.k, 0, nt)
    dt = time.time() - t0
    off, tg, vl = ix.arrays()
    t_arr = time.time() - t0 - dt
    print("scale=%d thr=%g targets=%d wall=%.3fs (%.0f targets/s) device=%.1fms entries=%d pops=%d edges=%d tier2=%d tier3=%d "
          "batch kernel %.1f ms x%d (+ %.3f s copying the index arrays to numpy)"
          % (a.scale, a.thr, nt, dt, nt / dt, st.total_ms, len(tg), st.pops, st.edge_pushes, st.rounds, st.dense_nodes,
             st.class_ms[4], st.class_launches[4], t_arr), flush=True)
    ix.close()
if a.cpu:
    from oracle import oracle as orc
    og = orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)
    t0 = time.time()
    for t in range(a.cpu):
        og.backward_push(t, 0.15, a.thr, orc.FIFO)
    dt = time.time() - t0
    print("cpu oracle FIFO: %d targets in %.2fs = %.1f targets/s (dense-array port, incl. O(n) clears)" % (a.cpu, dt, a.cpu / dt))
g.close()
