"""Exploratory timing of the engine phases on an R-MAT graph (developer tool, not a test)."""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
pkg.set_kernel_timing(True)  # class times wanted here

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=20)
ap.add_argument("--queries", type=int, default=4)
ap.add_argument("--rounds", type=str, default="0")
ap.add_argument("--dense-frac", type=float, default=0.0)
ap.add_argument("--topk", type=int, default=0)
ap.add_argument("--eps", type=float, default=0.5)
ap.add_argument("--relabel", type=int, default=0)
a = ap.parse_args()

t0 = time.time()
if a.relabel:
    src, dst = pkg.rmat_edges(a.scale, 16, 1)
    n = 1 << a.scale
    deg = np.bincount(src, minlength=n)
    order = np.argsort(-deg, kind="stable")          # new id -> old id
    old2new = np.empty(n, dtype=np.int32)
    old2new[order] = np.arange(n, dtype=np.int32)
    host = pkg.HostCsr(n, old2new[src], old2new[dst])
else:
    host = pkg.HostCsr.rmat(a.scale, 16, seed=1)
print("graph n=%d m=%d built in %.1fs; dead ends %d; max in-deg %d max out-deg %d" % (
    host.n, host.m, time.time() - t0, int((np.diff(host.out_rp) == 0).sum()), int(np.diff(host.in_rp).max()),
    int(np.diff(host.out_rp).max())), flush=True)
t0 = time.time()
g = pkg.Graph(host)
print("upload %.1fs" % (time.time() - t0), flush=True)
if a.dense_frac > 0:
    t = pkg.tuning_default()
    t.dense_frac = a.dense_frac
    g.set_tuning(t)
rng = np.random.default_rng(2)
srcs = [int(x) for x in rng.integers(0, host.n, size=a.queries)]
if a.relabel:
    srcs = [int(old2new[x]) for x in srcs]
for rounds in [int(x) for x in a.rounds.split(",")]:
    for s in srcs:
        t0 = time.time()
        if a.topk:
            nsel, ids, vals, _, st = g.fora_topk(s, a.eps, 0.15, a.topk, seed=3)
        else:
            _, st = g.fora_single_source(s, a.eps, 0.15, seed=3, n_rounds=rounds, fetch=False)
        wall = (time.time() - t0) * 1e3
        d = st.as_dict()
        print("src=%d deg=%d rounds=%d wall=%.2fms total=%.2f push=%.2f mc=%.2f sel=%.2f levels=%d dense=%d pops=%d edges=%d "
              "walks=%d steps=%d rsum=%.4g rmax=%.3g dense=%.1fus/lvl sparse=%.1fus/batch dom=%s %.3fms x%d %.1fGB/s" % (
                  s, host.out_rp[s + 1] - host.out_rp[s], d["rounds"], wall, d["total_ms"], d["push_ms"], d["mc_ms"],
                  d["select_ms"], d["levels"], d["dense_levels"], d["pops"], d["edge_pushes"], d["walks"],
                  d["walk_steps"], d["rsum"], d["rmax_final"], 1e3 * d["class_ms"][1] / max(1, d["class_launches"][1]),
                  1e3 * d["class_ms"][2] / max(1, d["class_launches"][2]), pkg.KERNEL_NAMES[d["dominant_kernel_id"]],
                  d["dominant_kernel_ms"], d["dominant_kernel_launches"],
                  d["dominant_kernel_bytes"] / max(d["dominant_kernel_ms"], 1e-9) / 1e6), flush=True)
