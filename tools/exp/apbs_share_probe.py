"""Developer probe: the dense tier's work sharing on a small graph with tiny chunks (every level is posted), with the
library's debug watchdog on.  PPRHIP_APBS_TIER=2 PPRHIP_APBS_CHUNK=16 PPRHIP_APBS_DEBUG=1 python tools/exp/apbs_share_probe.py"""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
from oracle import oracle as orc
orc.build()
host = pkg.HostCsr.rmat(12, 16, seed=1)
og = orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)
with pkg.Graph(host) as g:
    for (lo, hi), thr in (((0, 8), 2e-4), ((0, 64), 2e-4), ((1000, 1600), 1e-3)):
        t0 = time.time()
        ix, st = g.all_pair_backward(0.15, thr, -1, lo, hi)
        off, tg, vl = ix.arrays()
        ooff, otg, ovl = og.all_pair_backward(0.15, thr, -1, lo, hi, schedule=orc.SYNC)
        print("targets [%d, %d) thr %g: %.3f s, entries %d, identical %s, max diff %.2e" % (
            lo, hi, thr, time.time() - t0, len(tg), np.array_equal(off, ooff) and np.array_equal(tg, otg),
            float(np.max(np.abs(vl - ovl))) if len(vl) == len(ovl) else -1), flush=True)
