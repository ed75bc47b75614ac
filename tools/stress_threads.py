"""Developer stress test of the worker-thread paths (batched top-k, All-Pair tier 3, batched FORA with
PPRHIP_BATCH_THREADS=1): many short calls with varying batch sizes; results must repeat exactly."""
import importlib, os, sys, time
import numpy as np
import torch  # noqa: F401
os.environ["PPRHIP_BATCH_THREADS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
host = pkg.HostCsr.rmat(13, 16, seed=3)
g = pkg.Graph(host)
g.set_tuning(pkg.tuning_batch())
rng = np.random.default_rng(0)
t0 = time.time()
ref = {}
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 150):
    q = int(rng.integers(1, 40))
    srcs = rng.integers(0, host.n, size=q).astype(np.int32)
    ids, vals, _ = g.fora_batch_topk(srcs, 8, 0.5, 0.15, seed=5)
    out, _, _, _, pq, _ = g.fora_batch_single_source(srcs, 0.5, 0.15, seed=5, fetch=True, per_query=True)
    for i, s in enumerate(srcs):
        key = int(s)
        sig = (int(pq[i].walks), int(pq[i].levels))
        if key in ref:
            assert ref[key][0] == sig and np.max(np.abs(ref[key][1] - out[i])) < 1e-12, (it, key)
        else:
            ref[key] = (sig, out[i].copy())
    if it % 10 == 0:
        os.environ["PPRHIP_APBS_TIER"] = "3"
        lo = int(rng.integers(0, host.n - 64))
        ix, st = g.all_pair_backward(0.15, 1e-4, -1, lo, lo + 48)
        ix.close()
        del os.environ["PPRHIP_APBS_TIER"]
        print("iteration", it, "ok, %.1f s" % (time.time() - t0), flush=True)
g.close()
print("stress ok")
