"""FORA top-32 on bench.py's live sources: 16 in flight and one at a time (developer tool; run once per environment
setting, e.g. PPRHIP_SPARSE_WG=0 / 1, and compare on the same box)."""
import importlib
import os
import sys
import time

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    import torch  # noqa: F401
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    host = bench.load_host(pkg, 22)
    live = np.nonzero(np.diff(host.out_rp) > 0)[0]
    srcs = bench.live_draw(np.random.default_rng(2), live, (3, 128))
    g = pkg.Graph(host, device=0)
    g.fora_batch_topk(srcs[0][:16], bench.TOPK, bench.EPS, bench.ALPHA, seed=1)
    for rep in (1, 2):
        t0 = time.perf_counter()
        g.fora_batch_topk(srcs[rep], bench.TOPK, bench.EPS, bench.ALPHA, seed=5)
        dt = time.perf_counter() - t0
        print("16 in flight: %.1f queries/s" % (128 / dt), flush=True)
    t0 = time.perf_counter()
    for i, s in enumerate(srcs[1][:32]):
        g.fora_topk(int(s), bench.EPS, bench.ALPHA, bench.TOPK, seed=5 + i)
    dt = time.perf_counter() - t0
    print("one at a time: %.1f queries/s (%.2f ms)" % (32 / dt, 1e3 * dt / 32), flush=True)
    g.close()


if __name__ == "__main__":
    main()
