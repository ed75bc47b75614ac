#!/usr/bin/env python3
"""Records results of the HIP engine on the Game-of-Thrones graph for the world_size-2 gloo test (needs a GPU).

tests/test_sharding_gloo.py runs on CPU, where the engine cannot compute; so that what travels through the exchanges
there is the engine's own output and not the oracle's, this script records it once on an MI355X:
  * FORA top-5 rows (ids, values) of the test's 7 sources, query i with seed 100 + i (pprhip_fora_topk);
  * the All-Pair-Backward-Search shard index of each of 2 ranks (targets of pprhip_shard_target_range(r, 2, n),
    threshold 1e-3, k = -1: every entry, target order) from pprhip_all_pair_backward.
Output: tests/golden/got_engine_shards.npz (also copied to gpurun_out/ so that a gpurun call brings it back).

    python tests/golden/make_engine_fixture.py
"""
import importlib
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
A, K, WORLD = 0.15, 5, 2


def main():
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    got = pkg.HostCsr.from_neo4j_csv(os.path.join(HERE, "got", "GOT_Nodes.csv"), os.path.join(HERE, "got", "GOT_Rels.csv"))
    sources = np.random.default_rng(2).integers(0, got.n, size=7)
    out = {"sources": sources, "k": K, "world": WORLD}
    with pkg.Graph(got) as g:
        ids = np.full((len(sources), K), -1, dtype=np.int32)
        vals = np.zeros((len(sources), K))
        for i, s in enumerate(sources):
            n_sel, ti, tv, _, _ = g.fora_topk(int(s), 0.5, A, K, seed=100 + i, cap=K)
            ids[i, :len(ti)], vals[i, :len(tv)] = ti, tv
        out["topk_ids"], out["topk_vals"] = ids, vals
        for r in range(WORLD):
            lo, hi = pkg.shard_target_range(r, WORLD, got.n)
            ix, _ = g.all_pair_backward(A, 1e-3, -1, lo, hi)
            off, tg, vl = ix.arrays()
            out["off%d" % r], out["tg%d" % r], out["vl%d" % r] = off, tg, vl
            ix.close()
    path = os.path.join(HERE, "got_engine_shards.npz")
    np.savez(path, **out)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    shutil.copy(path, os.path.join(ROOT, "gpurun_out", "got_engine_shards.npz"))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
