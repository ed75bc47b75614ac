"""Single-source FORA one query at a time at R-MAT `scale` (the drop-in call): queries/s, dense levels, and the dense
class's time per level.   python tools/exp/single_rate.py [scale] [queries]"""
import importlib
import os
import sys
import time

import numpy as np
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
q = int(sys.argv[2]) if len(sys.argv) > 2 else 24
host = pkg.HostCsr.rmat(scale, 16, seed=1)
live = np.nonzero(np.diff(host.out_rp) > 0)[0].astype(np.int32)
srcs = live[np.random.default_rng(2).integers(0, live.size, size=q)]
with pkg.Graph(host) as g:
    for s in srcs[:2]:
        g.fora_single_source(int(s), 0.5, 0.15, seed=3)
    t0 = time.perf_counter()
    dl = 0
    for i, s in enumerate(srcs):
        _, st = g.fora_single_source(int(s), 0.5, 0.15, seed=3 + i)
        dl += st.dense_levels
    dt = time.perf_counter() - t0
    pkg.set_kernel_timing(True)
    ms = cnt = 0
    tot = 0.0
    for i, s in enumerate(srcs[:8]):
        _, st = g.fora_single_source(int(s), 0.5, 0.15, seed=3 + i)
        ms += st.class_ms[1]
        cnt += st.class_launches[1]
        tot += st.total_ms
    print("%s R-MAT %d: %.1f queries/s (%.2f ms each), %.1f dense levels per query; timed: %.1f us per dense level, %.2f ms device time per query"
          % (os.environ.get("TAG", ""), scale, q / dt, 1e3 * dt / q, dl / q, 1e3 * ms / max(1, cnt), tot / 8), flush=True)
