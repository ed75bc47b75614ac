"""world_size-2 `gloo` tests of the N > 1 paths (runs on CPU): query sharding + top-k gather
(config #4) and target-range sharding + index gather/merge (config #5).  No GPU exists here, so each
rank's compute is a recorded result of the HIP engine on an MI355X (tests/golden/got_engine_shards.npz,
written by tests/golden/make_engine_fixture.py: FORA top-5 rows and the All-Pair shard index of every
rank); what is under test is the sharding, the exchange and the product's merge, and the outcome is
compared with the unsharded CPU oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = 0.15
K = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import importlib
    import torch
    import torch.distributed as dist
    from conftest import GOT_NODES, GOT_RELS
    from oracle import oracle as orc
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    sh = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd.sharding")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    try:
        got = pkg.HostCsr.from_neo4j_csv(GOT_NODES, GOT_RELS)
        rec = np.load(os.path.join(ROOT, "tests", "golden", "got_engine_shards.npz"))  # the engine's results
        assert int(rec["world"]) == world and int(rec["k"]) == K
        # ---- batched FORA top-k: query i on rank i mod world, gather top-k blocks to rank 0
        sources = np.random.default_rng(2).integers(0, got.n, size=7)
        assert np.array_equal(sources, rec["sources"])
        idx, mine = sh.shard_sources(sources, rank, world)
        ids = [rec["topk_ids"][int(i)] for i in idx]
        vals = [rec["topk_vals"][int(i)] for i in idx]
        res = sh.gather_topk(dist, torch, ids, vals, len(sources), K, rank, world)
        # ---- All-Pair-Backward-Search: contiguous target ranges, gather shard arrays, merge on rank 0
        lo, hi = sh.target_range(rank, world, got.n)
        off, tg, vl = rec["off%d" % rank], rec["tg%d" % rank], rec["vl%d" % rank]
        # the exchange that scales: every rank ends with the merged lists of the sources it owns
        # (partitioned by the library's owner rule, finalised by the library: only the fabric is gloo)
        own = sh.exchange_index_by_source(dist, torch, off, tg, vl, rank, world, got.n, 3)
        o_off, o_tg, o_vl = own.arrays()
        np.savez(os.path.join(outdir, "own%d.npz" % rank), off=o_off, tg=o_tg, vl=o_vl, lo=lo, hi=hi)
        shards = sh.gather_index(dist, torch, off, tg, vl, rank, world)
        if rank == 0:
            parts = [pkg.index_from_arrays(got.n, o, t, v) for o, t, v in shards]
            m_off, m_tg, m_vl = pkg.merge_indexes(parts, 3).arrays()
            np.savez(os.path.join(outdir, "rank0.npz"), ids=res[0], vals=res[1], sources=sources, off=m_off, tg=m_tg,
                     vl=m_vl)
        else:
            assert res is None and shards is None
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo(tmp_path, orc, got):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    d = np.load(tmp_path / "rank0.npz")
    from conftest import to_oracle
    og = to_oracle(orc, got)
    # every query's row equals the single-process result, in query order
    for i, s in enumerate(d["sources"]):
        est, _ = og.fora_topk(int(s), 0.5, A, K, seed=100 + i, schedule=orc.SYNC)
        cnt, ti, tv = orc.topk(est, K, cap=K)
        assert list(d["ids"][i][:len(ti)]) == list(ti) and np.max(np.abs(d["vals"][i][:len(tv)] - tv), initial=0) <= 1e-9
        assert np.all(d["ids"][i][len(ti):] == -1)
    off, tg, vl = og.all_pair_backward(A, 1e-3, 3)
    assert np.array_equal(d["off"], off) and np.array_equal(d["tg"], tg) and np.max(np.abs(d["vl"] - vl)) <= 1e-12
    # the all-to-all form: rank r holds exactly the rows of its own sources, equal to the unsharded result
    for r in range(2):
        o = np.load(tmp_path / ("own%d.npz" % r))
        lo, hi = int(o["lo"]), int(o["hi"])
        assert o["off"][lo] == 0 and o["off"][-1] == o["off"][hi]                # nothing outside [lo, hi)
        for v in range(lo, hi):
            a, b = int(o["off"][v]), int(o["off"][v + 1])
            c, e = int(off[v]), int(off[v + 1])
            assert np.array_equal(o["tg"][a:b], tg[c:e]) and np.max(np.abs(o["vl"][a:b] - vl[c:e]), initial=0) <= 1e-12


def test_owner_partition_is_the_range_rule(pkg):
    """pprhip_owner_partition (the host form of k_owner_partition's rule) sends source v to the rank whose
    pprhip_shard_target_range holds v, keeps the entries' order inside an owner's share, and rejects bad ids."""
    rng = np.random.default_rng(3)
    for n, w in ((107, 2), (4096, 3), (10, 10), (1000003, 8)):
        v = rng.integers(0, n, size=5000).astype(np.int32)
        counts, order = pkg.owner_partition(n, w, v)
        assert int(counts.sum()) == v.size and sorted(order.tolist()) == list(range(v.size))
        at = 0
        for r in range(w):
            lo, hi = pkg.shard_target_range(r, w, n)
            mine = order[at:at + int(counts[r])].astype(np.int64)
            assert np.all((v[mine] >= lo) & (v[mine] < hi)) and np.all(np.diff(mine) > 0)
            at += int(counts[r])
    with pytest.raises(pkg.PprhipError):
        pkg.owner_partition(10, 2, np.array([3, 10], dtype=np.int32))


def test_shard_helpers(pkg):
    import importlib
    sh = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd.sharding")
    for n in (1, 7, 8, 107, 1 << 20):
        for w in (1, 2, 3, 8):
            r = [sh.target_range(i, w, n) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 1
    src = np.arange(10) * 3
    seen = np.concatenate([sh.shard_sources(src, r, 4)[0] for r in range(4)])
    assert sorted(seen) == list(range(10))
