"""Writes the csr.bin tools/exp/apbs_levels.c reads: the R-MAT graph of the bench in the engine's internal vertex order
(nodes with in-edges first, then out-degree descending, stable: DESIGN.md 4).  python tools/exp/apbs_levels_csr.py 22 /tmp/r22.bin"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
scale, path = int(sys.argv[1]), sys.argv[2]
h = pkg.HostCsr.rmat(scale, 16, seed=1)
od = np.diff(h.out_rp).astype(np.int64)
ind = np.diff(h.in_rp).astype(np.int64)
key = np.where(ind > 0, 0, 1) * (1 << 40) + ((1 << 32) - od)
new2old = np.argsort(key, kind="stable")
old2new = np.empty(h.n, dtype=np.int64)
old2new[new2old] = np.arange(h.n)
deg_new = ind[new2old]
irp = np.zeros(h.n + 1, dtype=np.uint32)
irp[1:] = np.cumsum(deg_new)
starts = h.in_rp[:-1].astype(np.int64)[new2old]
idx = np.repeat(starts - irp[:-1].astype(np.int64), deg_new) + np.arange(h.m, dtype=np.int64)
ici = old2new[h.in_ci[idx]].astype(np.int32)
with open(path, "wb") as f:
    f.write(np.uint32(h.n).tobytes())
    f.write(np.uint64(h.m).tobytes())
    f.write(od[new2old].astype(np.uint32).tobytes())
    f.write(irp.tobytes())
    f.write(ici.tobytes())
print("wrote", path, h.n, h.m)
