"""Differential test on random small graphs (self loops, multi-edges, dead ends, isolated nodes, empty graphs):
every entry point of the path against the CPU oracle's frontier-synchronous twin, under the level shapes a tiny
graph would otherwise never reach (dense levels forced from the first edge on).  GPU, through the C ABI."""
import numpy as np
import pytest

from conftest import edges_to_host, to_oracle

pytestmark = pytest.mark.gpu
A = 0.15


def random_graph(pkg, seed):
    if seed == 0:
        return edges_to_host(pkg, 3, [])                                  # no edges at all
    if seed == 1:
        return edges_to_host(pkg, 2, [(0, 0), (0, 0), (1, 0)])            # self loops only + one edge into them
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(2, 90))
    m = int(rng.integers(0, 6 * n))
    src = rng.integers(0, n, size=m)
    dst = rng.integers(0, n, size=m)
    if seed % 3 == 0 and m:                      # a hub and a few dead ends
        dst[: m // 3] = int(rng.integers(0, n))
        src[src == (n - 1)] = 0
    return edges_to_host(pkg, n, list(zip(src.tolist(), dst.tolist())))


def orc_tuning(orc, t):
    o = orc.tuning_default()
    for f, _ in o._fields_:
        setattr(o, f, getattr(t, f))
    return o


@pytest.mark.parametrize("seed", range(12))
def test_random_graph_against_twin(pkg, orc, seed):
    host = random_graph(pkg, seed)
    og = to_oracle(orc, host)
    rng = np.random.default_rng(seed)
    srcs = sorted(set(int(x) for x in rng.integers(0, host.n, size=6)))
    with pkg.Graph(host) as g:
        for dense_frac in (1e-9, 0.05, 1e9):
            t = pkg.tuning_default()
            t.dense_frac = dense_frac
            g.set_tuning(t)
            ot = orc_tuning(orc, t)
            for s in srcs:
                for rmax in (1e-2, 1e-6):
                    p, r, rsum, st = g.forward_push(s, A, rmax)
                    po, ro, rso, sto = og.forward_push(s, A, rmax, orc.SYNC)
                    assert st.levels == sto.levels, (seed, dense_frac, s, rmax)
                    assert np.max(np.abs(p - po)) <= 1e-12 and np.max(np.abs(r - ro)) <= 1e-12
                    pb, rb, stb = g.backward_push(s, A, rmax)
                    pbo, rbo, stbo = og.backward_push(s, A, rmax, orc.SYNC)
                    assert stb.levels == stbo.levels
                    assert np.max(np.abs(pb - pbo)) <= 1e-12 and np.max(np.abs(rb - rbo)) <= 1e-12
                est, st = g.fora_single_source(s, 0.5, A, seed=7, n_rounds=0)
                ref, sto = og.fora_whole(s, 0.5, A, seed=7, n_rounds=0, schedule=orc.SYNC, tuning=ot)
                assert st.rounds == sto.rounds and st.walks == sto.walks
                assert np.max(np.abs(est - ref)) <= 1e-9
                k = 1 + s % 5
                nsel, ids, vals, _, _ = g.fora_topk(s, 0.5, A, k, seed=9, cap=host.n)
                reft, _ = og.fora_topk(s, 0.5, A, k, seed=9, schedule=orc.SYNC)
                cnt, oids, _ = orc.topk(reft, k, cap=host.n)
                assert nsel == cnt and list(ids) == list(oids)
            # all sources at once through the batched entry points
            out, _, _, _, pq, _ = g.fora_batch_single_source(srcs, 0.5, A, seed=7, fetch=True, per_query=True)
            for i, s in enumerate(srcs):
                ref, sto = og.fora_whole(s, 0.5, A, seed=7, n_rounds=0, schedule=orc.SYNC, tuning=ot)
                assert pq[i].walks == sto.walks and np.max(np.abs(out[i] - ref)) <= 1e-9
            ids, vals, _ = g.fora_batch_topk(srcs, 3, 0.5, A, seed=9)
            for i, s in enumerate(srcs):
                reft, _ = og.fora_topk(s, 0.5, A, 3, seed=9 + i, schedule=orc.SYNC)
                cnt, oids, _ = orc.topk(reft, 3, cap=3)
                m = min(cnt, 3)
                assert list(ids[i][:m]) == list(oids[:m]) and np.all(ids[i][m:] == -1)
        # All-Pair with every tier as the starting tier
        g.set_tuning(pkg.tuning_default())
        for thr, k in ((1e-2, -1), (1e-4, 3)):
            ooff, otg, ovl = og.all_pair_backward(A, thr, k, schedule=orc.SYNC)
            ix, _ = g.all_pair_backward(A, thr, k)
            off, tg, vl = ix.arrays()
            assert np.array_equal(off, ooff) and np.array_equal(tg, otg) and np.max(np.abs(vl - ovl), initial=0) <= 1e-12
            ix.close()


@pytest.mark.parametrize("tier", ["2", "3"])
def test_random_graph_all_pair_tiers(pkg, orc, tier, monkeypatch):
    monkeypatch.setenv("PPRHIP_APBS_TIER", tier)
    for seed in (3, 4, 5):
        host = random_graph(pkg, seed)
        og = to_oracle(orc, host)
        with pkg.Graph(host) as g:
            ooff, otg, ovl = og.all_pair_backward(A, 1e-4, 3, schedule=orc.SYNC)
            ix, _ = g.all_pair_backward(A, 1e-4, 3)
            off, tg, vl = ix.arrays()
            assert np.array_equal(off, ooff) and np.array_equal(tg, otg) and np.max(np.abs(vl - ovl), initial=0) <= 1e-12
            ix.close()
