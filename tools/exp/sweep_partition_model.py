"""Would partitioning the batched sweep's edges by source range over the eight XCDs pay?  
Counts, on the R-MAT graph in the engine's internal vertex order (nodes with in-edges first, then out-degree
descending): the in-edges by popularity tier of their source, and the (row, partition) segments - each one a
128-byte partial row sum that has to be written and read back - that an 8-way partition of the tier between "fits
every L2 today" (32 K lines) and "fits the eight L2s together" (256 K lines) would create.
    python tools/exp/sweep_partition_model.py 22"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
h = pkg.HostCsr.rmat(scale, 16, seed=1)
n, m = h.n, h.m
outdeg = np.diff(h.out_rp).astype(np.int64)
indeg = np.diff(h.in_rp).astype(np.int64)
no_in = (indeg == 0)
order = np.lexsort((np.arange(n), -outdeg, no_in))      # engine: in-edge nodes first, out-degree descending, stable
old2new = np.empty(n, dtype=np.int64); old2new[order] = np.arange(n)
src_new = old2new[h.in_ci[:m]]                          # source (internal id) of every in-edge, row-major
row_of = np.repeat(np.arange(n), indeg)                 # destination row (original id) of every in-edge
print("n=%d m=%d rows with in-edges %d, nodes with out-edges %d" % (n, m, int((indeg > 0).sum()), int((outdeg > 0).sum())))
tiers = [("LDS table (1 K lines)", 0, 1024), ("every L2 today (to 32 K)", 1024, 32768),
         ("eight L2s together (32 K - 256 K)", 32768, 262144), ("beyond (256 K -)", 262144, n)]
for name, lo, hi in tiers:
    sel = (src_new >= lo) & (src_new < hi)
    print("%-36s %6.2f %% of the in-edges (%d)" % (name, 100.0 * sel.sum() / m, int(sel.sum())))
sel = (src_new >= 32768) & (src_new < 262144)
part = (src_new[sel] - 32768) // ((262144 - 32768) // 8)
seg = np.unique(row_of[sel] * 8 + np.minimum(part, 7)).size
print("8-way partition of the 32 K - 256 K tier: %d edges -> %d (row, partition) segments = %.2f per edge" % (int(sel.sum()), seg, seg / sel.sum()))
print("  saved at best: %d L2 misses (every edge of the tier but the %d compulsory ones);" % (int(sel.sum()) - 229376, 229376))
print("  cost: %d partial-sum lines written + read = %d line transfers" % (seg, 2 * seg))
