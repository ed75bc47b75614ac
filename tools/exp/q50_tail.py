"""50-query calls (config #4's call shape) with the leftover rule on and off (developer tool, round 5).

    python3 tools/exp/q50_tail.py            -> queries/s of six 50-query calls, PPRHIP_BATCH_NO_TAIL unset / set

The rule (fora.cpp: kTailSingle) runs a call's last q % 16 <= 3 queries one at a time on the handle's own workspace
instead of as a last round of sweeps with nearly empty columns.  With the workspace pool there are no rounds of 16 any
more: this measures whether the rule still pays.
"""
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
    rng = np.random.default_rng(7)
    g = pkg.Graph(host, device=0)
    g.set_tuning(pkg.tuning_batch())
    conf = pkg.conf_whole_graph(host.n, host.m, bench.ALPHA)
    for q in (50, 51, 35, 20):
        srcs = bench.live_draw(rng, live, (7, q))
        store = pkg.Results(g, q)
        for tag, env in (("rule on ", None), ("rule off", "1"), ("rule on ", None), ("rule off", "1")):
            if env:
                os.environ["PPRHIP_BATCH_NO_TAIL"] = env
            else:
                os.environ.pop("PPRHIP_BATCH_NO_TAIL", None)
            g.fora_batch_single_source(srcs[0], bench.EPS, bench.ALPHA, seed=21, k=bench.TOPK, conf=conf, keep=store)
            t0 = time.perf_counter()
            for i in range(1, 7):
                g.fora_batch_single_source(srcs[i], bench.EPS, bench.ALPHA, seed=21 + i, k=bench.TOPK, conf=conf, keep=store)
            dt = time.perf_counter() - t0
            print("q = %d  %s  %7.1f queries/s  %6.1f ms per call" % (q, tag, 6 * q / dt, 1e3 * dt / 6), flush=True)
        store.close()
    g.close()


if __name__ == "__main__":
    main()
