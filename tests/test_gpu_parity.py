"""GPU parity: the HIP engine (through the C ABI) against the CPU oracle's frontier-synchronous twin
on the same seeded inputs.

Tolerances: `north_star` asks for reserve vectors within 1e-6 L-inf of the reference path and
identical top-k sets.  The engine differs from the oracle twin only in fp64 addition order, so
the tests hold it to 1e-12 where no random walk is involved and to 1e-9 where walk increments are
summed in a different order; TOL_SPEC = 1e-6 is asserted as the contractual bar as well.
"""
import numpy as np
import pytest

from conftest import shared_graph, to_oracle

pytestmark = pytest.mark.gpu

ALPHA = 0.15
TOL_SPEC = 1e-6
TOL_PUSH = 1e-12
TOL_MC = 1e-9


@pytest.fixture
def dev_got(pkg, got, dev_cache):
    return shared_graph(dev_cache, pkg, "dev_got", lambda: pkg.Graph(got))


@pytest.fixture
def dev_rmat12(pkg, rmat12, dev_cache):
    return shared_graph(dev_cache, pkg, "dev_rmat12", lambda: pkg.Graph(rmat12))


@pytest.fixture
def dev_rmat15(pkg, rmat15, dev_cache):
    return shared_graph(dev_cache, pkg, "dev_rmat15", lambda: pkg.Graph(rmat15))


def assert_close(a, b, tol, what):
    err = float(np.max(np.abs(a - b))) if a.size else 0.0
    assert err <= tol, "%s: max abs diff %.3e > %.1e" % (what, err, tol)
    assert err <= TOL_SPEC


def to_orc_tuning(orc, t):
    o = orc.tuning_default()
    for f, _ in o._fields_:
        setattr(o, f, getattr(t, f))
    return o


def sources(host, count, seed=2):
    rng = np.random.default_rng(seed)
    return [int(x) for x in rng.integers(0, host.n, size=count)]


# ------------------------------------------------------------------ forward push (a1)
@pytest.mark.parametrize("rmax", [7.554e-4, 1e-5, 1e-8])
def test_forward_push_got(pkg, orc, got, dev_got, rmax):
    og = to_oracle(orc, got)
    for s in [0, 5, 17, 42, 99, 106] + sources(got, 6):
        p, r, rsum, st = dev_got.forward_push(s, ALPHA, rmax)
        po, ro, rsum_o, sto = og.forward_push(s, ALPHA, rmax, orc.SYNC)
        assert_close(p, po, TOL_PUSH, "reserve src=%d" % s)
        assert_close(r, ro, TOL_PUSH, "residue src=%d" % s)
        assert abs(rsum - rsum_o) <= 1e-12
        assert st.levels == sto.levels
        assert st.pops + st.dense_nodes == sto.pops + sto.dense_nodes
        assert st.dead_end_pops == sto.dead_end_pops
        assert st.enqueues == sto.enqueues


def test_forward_push_toys(pkg, orc, toy_graphs):
    for name, host in toy_graphs.items():
        og = to_oracle(orc, host)
        with pkg.Graph(host) as g:
            for s in range(host.n):
                for rmax in (1e-2, 1e-6, 1e-12):
                    p, r, rsum, st = g.forward_push(s, ALPHA, rmax)
                    po, ro, rsum_o, sto = og.forward_push(s, ALPHA, rmax, orc.SYNC)
                    assert_close(p, po, TOL_PUSH, "%s reserve src=%d" % (name, s))
                    assert_close(r, ro, TOL_PUSH, "%s residue src=%d" % (name, s))


@pytest.mark.parametrize("dense_frac", [0.05, 1e-9, 1e9])
def test_forward_push_rmat12_modes(pkg, orc, rmat12, dev_rmat12, dense_frac):
    """Sparse-only, dense-only and mixed level shapes give the same vectors."""
    og = to_oracle(orc, rmat12)
    t = pkg.tuning_default()
    t.dense_frac = dense_frac
    dev_rmat12.set_tuning(t)
    try:
        for s in sources(rmat12, 4):
            for rmax in (1e-4, 1e-7):
                p, r, rsum, st = dev_rmat12.forward_push(s, ALPHA, rmax)
                po, ro, rsum_o, sto = og.forward_push(s, ALPHA, rmax, orc.SYNC)
                assert_close(p, po, TOL_PUSH, "reserve src=%d rmax=%g" % (s, rmax))
                assert_close(r, ro, TOL_PUSH, "residue src=%d rmax=%g" % (s, rmax))
                assert st.levels == sto.levels
                if dense_frac == 1e9 and po.sum() < 1.0:
                    assert st.dense_levels == 0
                if dense_frac == 1e-9 and st.levels:
                    assert st.dense_levels == st.levels
    finally:
        dev_rmat12.set_tuning(pkg.tuning_default())


def test_forward_push_rmat15(pkg, orc, rmat15, dev_rmat15):
    og = to_oracle(orc, rmat15)
    for s in sources(rmat15, 3):
        p, r, rsum, st = dev_rmat15.forward_push(s, ALPHA, 1e-7)
        po, ro, rsum_o, sto = og.forward_push(s, ALPHA, 1e-7, orc.SYNC)
        assert_close(p, po, TOL_PUSH, "reserve src=%d" % s)
        assert_close(r, ro, TOL_PUSH, "residue src=%d" % s)
        # push invariant: reserve + residue mass is conserved
        assert abs(p.sum() + r.sum() - 1.0) < 1e-12 or rmat15.out_rp[s + 1] == rmat15.out_rp[s]


# ------------------------------------------------------------------ random walks (a3, a4)
@pytest.mark.parametrize("nzh", [False, True])
def test_walk_terminals_bit_exact(pkg, orc, got, dev_got, rmat12, dev_rmat12, nzh):
    for host, dev in ((got, dev_got), (rmat12, dev_rmat12)):
        og = to_oracle(orc, host)
        rng = np.random.default_rng(7)
        starts = rng.integers(0, host.n, size=4000).astype(np.int32)
        idx = rng.integers(0, 1 << 40, size=4000).astype(np.uint64)
        term, steps = dev.random_walks(starts, idx, ALPHA, seed=3, stream=5, no_zero_hop=nzh)
        for i in range(starts.size):
            t, st = og.random_walk(int(starts[i]), ALPHA, 3, 5, int(idx[i]), nzh)
            assert t == term[i] and st == steps[i], "walk %d differs" % i


# ------------------------------------------------------------------ FORA whole graph (a5)
@pytest.mark.parametrize("n_rounds", [1, 2, 4, 0])
def test_fora_single_source_got(pkg, orc, got, dev_got, n_rounds):
    og = to_oracle(orc, got)
    for s in [0, 17, 42, 106] + sources(got, 4, seed=11):
        est, st = dev_got.fora_single_source(s, 0.5, ALPHA, seed=3, n_rounds=n_rounds)
        ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=3, n_rounds=n_rounds, schedule=orc.SYNC)
        assert st.rounds == sto.rounds
        assert st.walks == sto.walks and st.walk_steps == sto.walk_steps
        assert_close(est, ref, TOL_MC, "fora src=%d" % s)
        # all mass is delivered unless floor(omega * rsum) == 0 (then the reference drops (1-alpha)*rsum too)
        assert abs(est.sum() - 1.0) < 1e-9 or (st.walks == 0 and abs(est.sum() + st.rsum - 1.0) < 1e-9)


def test_fora_single_source_rmat12(pkg, orc, rmat12, dev_rmat12):
    og = to_oracle(orc, rmat12)
    for s in sources(rmat12, 3, seed=5):
        for n_rounds in (1, 3):
            est, st = dev_rmat12.fora_single_source(s, 0.5, ALPHA, seed=9, n_rounds=n_rounds)
            ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=9, n_rounds=n_rounds, schedule=orc.SYNC)
            assert st.walks == sto.walks and st.walk_steps == sto.walk_steps
            assert_close(est, ref, TOL_MC, "fora src=%d rounds=%d" % (s, n_rounds))


def test_fora_auto_rounds_match_twin(pkg, orc, rmat12, dev_rmat12):
    og = to_oracle(orc, rmat12)
    s = sources(rmat12, 1, seed=21)[0]
    est, st = dev_rmat12.fora_single_source(s, 0.5, ALPHA, seed=1, n_rounds=0)
    ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=1, n_rounds=0, schedule=orc.SYNC)
    assert st.rounds == sto.rounds >= 1
    assert_close(est, ref, TOL_MC, "fora auto")


# ------------------------------------------------------------------ batched FORA (config #4 shape, a5 per query)
@pytest.mark.parametrize("dense_frac", [0.05, 1e-9, 1e9])
def test_fora_batch_got(pkg, orc, got, dev_got, dense_frac):
    """Queries in flight together give what each gives alone: against the twin and against the single-query
    entry point, for every level shape (the batched dense sweep serves up to 8 queries at once)."""
    og = to_oracle(orc, got)
    t = pkg.tuning_default()
    t.dense_frac = dense_frac
    dev_got.set_tuning(t)
    try:
        srcs = [0, 17, 42, 106, 90, 3] + sources(got, 13, seed=19)      # 19 queries: more than two rounds of slots
        for n_rounds in (1, 3, 0):
            out, ids, vals, nsel, pq, st = dev_got.fora_batch_single_source(srcs, 0.5, ALPHA, seed=3, n_rounds=n_rounds,
                                                                            k=10, fetch=True, per_query=True)
            assert st.levels == sum(x.levels for x in pq)
            for i, s in enumerate(srcs):
                ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=3, n_rounds=n_rounds, schedule=orc.SYNC,
                                         tuning=to_orc_tuning(orc, t))   # level shapes decide where a round is cut
                assert pq[i].rounds == sto.rounds and pq[i].levels == sto.levels
                assert pq[i].walks == sto.walks and pq[i].walk_steps == sto.walk_steps
                assert_close(out[i], ref, TOL_MC, "batch src=%d" % s)
                single, sts = dev_got.fora_single_source(s, 0.5, ALPHA, seed=3, n_rounds=n_rounds)
                assert sts.walks == pq[i].walks and sts.dense_levels == pq[i].dense_levels
                assert_close(out[i], single, TOL_PUSH, "batch vs single src=%d" % s)
                cnt, oids, ovals = orc.topk(out[i], 10, cap=10)
                assert nsel[i] == cnt
                m = min(cnt, 10)
                assert list(ids[i][:m]) == list(oids[:m]) and np.array_equal(vals[i][:m], ovals[:m])
                assert np.all(ids[i][m:] == -1) and np.all(vals[i][m:] == 0.0)
    finally:
        dev_got.set_tuning(pkg.tuning_default())


@pytest.mark.parametrize("threads", ["0", "1", "0-walks-in-line"])
def test_fora_batch_rmat12(pkg, orc, rmat12, threads, monkeypatch):
    """Batch profile of the cost model on both sides; one worker thread per slot (1) or all slots on the caller (0),
    there with the queries' walk phases on a side stream beside the sweeps (default) or in line with everything else."""
    monkeypatch.setenv("PPRHIP_BATCH_THREADS", threads[0])
    if threads.endswith("in-line"):
        monkeypatch.setenv("PPRHIP_BATCH_WALKS_BESIDE", "0")  # read when the handle's first batch starts
    og = to_oracle(orc, rmat12)
    srcs = sources(rmat12, 21, seed=8)
    t = pkg.tuning_batch()
    dev_rmat12 = pkg.Graph(rmat12)
    dev_rmat12.set_tuning(t)
    try:
        for n_rounds in (2, 0):
            out, _, _, _, pq, st = dev_rmat12.fora_batch_single_source(srcs, 0.5, ALPHA, seed=9, n_rounds=n_rounds,
                                                                       fetch=True, per_query=True)
            assert st.dense_levels > 0 and st.class_launches[5] > 0      # the batched sweep ran
            assert st.class_launches[5] < st.dense_levels                 # and served several queries per launch
            for i, s in enumerate(srcs):
                ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=9, n_rounds=n_rounds, schedule=orc.SYNC,
                                         tuning=to_orc_tuning(orc, t))
                assert pq[i].rounds == sto.rounds and pq[i].walks == sto.walks and pq[i].levels == sto.levels
                assert_close(out[i], ref, TOL_MC, "batch src=%d" % s)
        dev_rmat12.set_tuning(pkg.tuning_default())
        # an empty batch and a batch of dead-end sources only
        out, _, _, _, _, st = dev_rmat12.fora_batch_single_source([], 0.5, ALPHA, seed=1, fetch=True)
        assert out.shape == (0, rmat12.n) and st.levels == 0
        dead = [int(v) for v in np.nonzero(np.diff(rmat12.out_rp) == 0)[0][:3]]
        if dead:
            out, _, _, _, _, st = dev_rmat12.fora_batch_single_source(dead, 0.5, ALPHA, seed=1, fetch=True)
            for i, s in enumerate(dead):
                assert out[i][s] == 1.0 and out[i].sum() == 1.0
    finally:
        dev_rmat12.close()


@pytest.mark.parametrize("q", [18, 35])
def test_fora_batch_leftover_queries_run_singly(pkg, orc, rmat15, dev_rmat15, q, monkeypatch):
    """With one workspace per column (PPRHIP_BATCH_WORKSPACES=16: a device without room for the pool) a call whose query
    count leaves one to three queries over after the full rounds of 16 (PPR.java:179's 50 = 3 x 16 + 2) runs those on the
    handle's own workspace, one at a time (fora.cpp: kTailSingle): every query - the leftovers too - equals the twin
    and what the call gives with the leftovers on the slots (PPRHIP_BATCH_NO_TAIL) and with the pool (the default, which
    has no such rule), vectors kept in a store and delivered to the host alike, top-k per query included."""
    og = to_oracle(orc, rmat15)
    srcs = sources(rmat15, q, seed=14)
    t = pkg.tuning_batch()
    dev_rmat15.set_tuning(t)
    store = pkg.Results(dev_rmat15, q)
    try:
        out3, ids3, vals3, nsel3, pq3, _ = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=5, k=8, fetch=True,
                                                                               per_query=True)
        monkeypatch.setenv("PPRHIP_BATCH_WORKSPACES", "16")
        out, ids, vals, nsel, pq, st = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=5, k=8, fetch=True,
                                                                           per_query=True, keep=store)
        assert np.max(np.abs(out - out3)) <= 1e-9 and np.array_equal(nsel, nsel3)
        assert all(pq[i].walks == pq3[i].walks and pq[i].levels == pq3[i].levels for i in range(q))
        monkeypatch.setenv("PPRHIP_BATCH_NO_TAIL", "1")
        out2, ids2, vals2, nsel2, pq2, _ = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=5, k=8, fetch=True,
                                                                               per_query=True)
        assert np.max(np.abs(out - out2)) <= 1e-9 and np.array_equal(nsel, nsel2)
        for i in list(range(q - 4, q)) + [0, 16]:
            assert pq[i].walks == pq2[i].walks and pq[i].levels == pq2[i].levels and pq[i].rounds == pq2[i].rounds
            ref, sto = og.fora_whole(srcs[i], 0.5, ALPHA, seed=5, n_rounds=0, schedule=orc.SYNC,
                                     tuning=to_orc_tuning(orc, t))
            assert pq[i].walks == sto.walks and pq[i].levels == sto.levels
            assert_close(out[i], ref, TOL_MC, "query %d src=%d" % (i, srcs[i]))
            assert np.max(np.abs(store.fetch(i) - out[i])) == 0.0
            cnt, oids, ovals = orc.topk(out[i], 8, cap=8)
            m = min(cnt, 8)
            assert nsel[i] == cnt and list(ids[i][:m]) == list(oids[:m]) and np.array_equal(vals[i][:m], ovals[:m])
    finally:
        store.close()
        dev_rmat15.set_tuning(pkg.tuning_default())


def test_fora_batch_rmat15_many_queries(pkg, orc, rmat15, dev_rmat15):
    """More queries than slots on a graph whose pushes run many dense levels; a sample is checked against the twin,
    every query against mass conservation, and a second call on the same handle gives the same vectors."""
    og = to_oracle(orc, rmat15)
    srcs = sources(rmat15, 40, seed=4)
    t = pkg.tuning_batch()
    dev_rmat15.set_tuning(t)
    try:
        out, ids, vals, nsel, pq, st = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=5, k=16, fetch=True,
                                                                           per_query=True)
        out2, _, _, _, _, _ = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=5, fetch=True)
        assert np.max(np.abs(out - out2)) < 1e-12
        for i, s in enumerate(srcs):
            assert abs(out[i].sum() - 1.0) < 1e-9 and out[i].min() >= 0.0
            cnt, oids, ovals = orc.topk(out[i], 16, cap=16)
            m = min(cnt, 16)
            assert nsel[i] == cnt and list(ids[i][:m]) == list(oids[:m]) and np.array_equal(vals[i][:m], ovals[:m])
        for i in (0, 7, 23, 39):
            ref, sto = og.fora_whole(srcs[i], 0.5, ALPHA, seed=5, n_rounds=0, schedule=orc.SYNC,
                                     tuning=to_orc_tuning(orc, t))
            assert pq[i].rounds == sto.rounds and pq[i].walks == sto.walks
            assert_close(out[i], ref, TOL_MC, "batch src=%d" % srcs[i])
    finally:
        dev_rmat15.set_tuning(pkg.tuning_default())


@pytest.mark.parametrize("env", [{"PPRHIP_BATCH_WORKSPACES": "16"}, {"PPRHIP_BATCH_WORKSPACES": "19"},
                                 {"PPRHIP_BATCH_WORKSPACES": "48"}, {"PPRHIP_BATCH_SLOTS_BESIDE": "0"},
                                 {"PPRHIP_BATCH_NO_HOOK": "1"}, {"PPRHIP_BATCH_NO_EARLY": "1", "PPRHIP_BATCH_WALKS_BESIDE": "0"}])
def test_batch_driver_variants(pkg, orc, rmat15, env, monkeypatch):
    """The sequential batch driver of round 5 (fora.cpp: SlotDriver): sweeps launched and collected separately, the
    queries outside a sweep stepped on a second stream beside it, a pool of workspaces (default 32) for the 16 columns
    of the contribution array, turns taken from inside a workspace's read-back wait.  Each query is the single-query
    algorithm whatever the driver does around it: with one workspace per column, an odd number of them, three per
    column, the workspaces on the sweeps' own stream, no turns from inside waits, no early first sweep and the walks in
    line, every query has the levels, rounds and walks of the default configuration and its vector to 1e-12 (sums that
    cross a chunk are atomic), with threshold rounds fixed at 3 (round starts of workspaces that hold no column list
    their start set) and chosen by the cost model; a sample is held to the twin."""
    og = to_oracle(orc, rmat15)
    srcs = sources(rmat15, 45, seed=31)
    t = pkg.tuning_batch()
    ref_g = pkg.Graph(rmat15)
    ref_g.set_tuning(t)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = pkg.Graph(rmat15)
    g.set_tuning(t)
    try:
        for n_rounds in (3, 0):
            out, _, _, _, pq, st = g.fora_batch_single_source(srcs, 0.5, ALPHA, seed=6, n_rounds=n_rounds, fetch=True,
                                                              per_query=True)
            for k in env:
                monkeypatch.delenv(k)
            out0, _, _, _, pq0, st0 = ref_g.fora_batch_single_source(srcs, 0.5, ALPHA, seed=6, n_rounds=n_rounds,
                                                                     fetch=True, per_query=True)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            assert st.dense_levels == st0.dense_levels and st.class_launches[5] > 0
            assert np.max(np.abs(out - out0)) <= 1e-12
            for i in range(len(srcs)):
                assert (pq[i].levels, pq[i].rounds, pq[i].walks, pq[i].dense_levels) == \
                       (pq0[i].levels, pq0[i].rounds, pq0[i].walks, pq0[i].dense_levels)
            for i in (0, 17, 44):
                ref, sto = og.fora_whole(srcs[i], 0.5, ALPHA, seed=6, n_rounds=n_rounds, schedule=orc.SYNC,
                                         tuning=to_orc_tuning(orc, t))
                assert pq[i].rounds == sto.rounds and pq[i].walks == sto.walks and pq[i].levels == sto.levels
                assert_close(out[i], ref, TOL_MC, "batch src=%d" % srcs[i])
    finally:
        g.close()
        ref_g.close()


@pytest.mark.parametrize("graph,relabel", [("got", "1"), ("rmat12", "1"), ("rmat15", "1"), ("rmat15", "0")])
def test_row_panel_single_query_sweep(pkg, orc, got, rmat12, rmat15, graph, relabel, monkeypatch):
    """The single-query dense level over the row-panel copy of the in-CSR (round 6: k_dense_edges_panel - a workgroup
    sums an item of one panel's source-sorted edges into accumulators in LDS, the apply kernel adds a row's parts; the
    layout of graphs from 2^20 edges on, forced here with PPRHIP_SWEEP1_PANELS=1): forward push with Jacobi and 2- /
    3-block Gauss-Seidel sweeps (whose bounds cut the one panel these graphs have), the power method, whole-graph FORA
    and FORA top-k, level for level and value for value the twin's (with PPRHIP_RELABEL=0: the other layout's) - on a
    handle that has run a query before (stale partial sums)."""
    host = {"got": got, "rmat12": rmat12, "rmat15": rmat15}[graph]
    og = to_oracle(orc, host)
    od = np.diff(host.out_rp).astype(np.float64)
    monkeypatch.setenv("PPRHIP_RELABEL", relabel)
    monkeypatch.setenv("PPRHIP_SWEEP1_PANELS", "1")
    g_part = pkg.Graph(host)
    monkeypatch.setenv("PPRHIP_SWEEP1_PANELS", "0")
    g_ref = pkg.Graph(host)
    srcs = [s for s in ([0, 17, 42] if graph == "got" else []) + list(sources(host, 8, seed=27)) if od[s] > 0][:4]
    try:
        for B in (1, 2, 3, 6):
            t = pkg.tuning_batch()       # (dense levels from 2 % of m)
            t.gs_blocks = B
            if graph == "got":
                t.dense_frac = 0.01
            g_part.set_tuning(t)
            g_ref.set_tuning(t)
            for s in srcs:
                p, r, rsum, st = g_part.forward_push(s, ALPHA, 1e-8)
                p0, r0, _, st0 = g_ref.forward_push(s, ALPHA, 1e-8)
                assert (st.dense_levels > 0 or graph == "got") and st.levels == st0.levels and st.dense_levels == st0.dense_levels
                assert_close(p, p0, TOL_PUSH, "reserve vs the other layout B=%d src=%d" % (B, s))
                assert_close(r, r0, TOL_PUSH, "residue vs the other layout B=%d src=%d" % (B, s))
                if relabel == "1":
                    po, ro, _, sto = og.forward_push(s, ALPHA, 1e-8, orc.SYNC)
                    assert st.levels == sto.levels and st.dense_levels == sto.dense_levels
                    assert_close(p, po, TOL_PUSH, "reserve B=%d src=%d" % (B, s))
                    assert_close(r, ro, TOL_PUSH, "residue B=%d src=%d" % (B, s))
                est, stf = g_part.fora_single_source(s, 0.5, ALPHA, seed=3)
                est0, stf0 = g_ref.fora_single_source(s, 0.5, ALPHA, seed=3)
                assert stf.walks == stf0.walks and stf.levels == stf0.levels
                assert_close(est, est0, TOL_MC, "FORA vs the other layout B=%d src=%d" % (B, s))
        g_part.set_tuning(pkg.tuning_default())
        g_ref.set_tuning(pkg.tuning_default())
        for s in srcs[:2]:
            pm, _ = g_part.power_method(s, ALPHA, 30)
            assert_close(pm, og.power_method(s, ALPHA, 30), TOL_PUSH, "power method src=%d" % s)
            nsel, ids, vals, est, st = g_part.fora_topk(s, 0.5, ALPHA, 10, seed=4, cap=host.n, fetch=True)
            nsel0, ids0, vals0, est0, st0 = g_ref.fora_topk(s, 0.5, ALPHA, 10, seed=4, cap=host.n, fetch=True)
            assert nsel == nsel0 and list(ids) == list(ids0) and st.rounds == st0.rounds
            assert_close(est, est0, TOL_MC, "top-k estimate src=%d" % s)
    finally:
        g_part.close()
        g_ref.close()


# ------------------------------------------------------------------ FORA top-k (a6, a7)
@pytest.mark.parametrize("k", [1, 10, 50, 200])
def test_fora_topk_got(pkg, orc, got, dev_got, k):
    og = to_oracle(orc, got)
    for s in [0, 17, 42, 106]:
        nsel, ids, vals, est, st = dev_got.fora_topk(s, 0.5, ALPHA, k, seed=4, cap=got.n, fetch=True)
        ref, sto = og.fora_topk(s, 0.5, ALPHA, k, seed=4, schedule=orc.SYNC)
        assert st.rounds == sto.rounds
        assert_close(est, ref, TOL_MC, "topk est src=%d" % s)
        cnt, oids, ovals = orc.topk(ref, k)
        assert nsel == cnt
        assert list(ids) == list(oids)  # identical top-k sets, identical order
        assert np.max(np.abs(vals - ovals)) <= TOL_MC if cnt else True


@pytest.mark.parametrize("dense_frac", [0.02, 0.05, 0.1, 0.3])
def test_fora_topk_got_mixed_level_shapes(pkg, orc, got, dev_got, dense_frac):
    """Every switch point between sparse and dense levels gives the twin's vectors (regression: a dense level
    that prepares only dead-end nodes hands the sparse shape an empty list but pending source mass)."""
    og = to_oracle(orc, got)
    t = pkg.tuning_default()
    t.dense_frac = dense_frac
    dev_got.set_tuning(t)
    try:
        for k in (10, 50):
            for s in (0, 17, 42, 99, 106):
                nsel, ids, vals, est, st = dev_got.fora_topk(s, 0.5, ALPHA, k, seed=4, cap=got.n, fetch=True)
                ref, sto = og.fora_topk(s, 0.5, ALPHA, k, seed=4, schedule=orc.SYNC)
                assert st.levels == sto.levels and st.walks == sto.walks
                assert_close(est, ref, TOL_MC, "topk est src=%d k=%d" % (s, k))
    finally:
        dev_got.set_tuning(pkg.tuning_default())


@pytest.mark.parametrize("threads", ["0", "1"])
def test_fora_batch_topk(pkg, orc, got, dev_got, rmat12, dev_rmat12, threads, monkeypatch):
    """Top-k queries in flight together: query i equals pprhip_fora_topk(src_i, seed + i) and the twin's top-k."""
    monkeypatch.setenv("PPRHIP_BATCH_THREADS", threads)
    for host, dev, k, srcs, tun in ((got, dev_got, 10, [0, 17, 42, 90, 106] + sources(got, 14, seed=3), pkg.tuning_default()),
                                    (rmat12, dev_rmat12, 32, sources(rmat12, 19, seed=6), pkg.tuning_batch())):
        og = to_oracle(orc, host)
        dev.set_tuning(tun)
        try:
            ids, vals, st = dev.fora_batch_topk(srcs, k, 0.5, ALPHA, seed=11)
            assert ids.shape == (len(srcs), k)
            for i, s in enumerate(srcs):
                nsel, sids, svals, _, sst = dev.fora_topk(s, 0.5, ALPHA, k, seed=11 + i, cap=k)
                m = min(nsel, k)
                assert list(ids[i][:m]) == list(sids[:m]) and np.max(np.abs(vals[i][:m] - svals[:m]), initial=0) < 1e-12
                assert np.all(ids[i][m:] == -1) and np.all(vals[i][m:] == 0.0)
                ref, sto = og.fora_topk(s, 0.5, ALPHA, k, seed=11 + i, schedule=orc.SYNC)
                cnt, oids, ovals = orc.topk(ref, k, cap=k)
                assert min(cnt, k) == m and list(oids[:m]) == list(ids[i][:m])
            assert st.levels > 0 and st.rounds >= len(srcs) - int(np.sum(np.diff(host.out_rp)[srcs] == 0))
        finally:
            dev.set_tuning(pkg.tuning_default())


@pytest.mark.parametrize("ahead", ["1", "0"])
def test_fora_topk_rmat12(pkg, orc, rmat12, dev_rmat12, ahead, monkeypatch):
    """ahead = 1 (default): the next round's push, sum and plan run on a second stream beside this round's walks and
    join the query only when the round is needed (engine.cpp: pprhip_fora_topk); 0: the rounds run one after another.
    Same rounds, walks, level counters and lists either way."""
    monkeypatch.setenv("PPRHIP_TOPK_AHEAD", ahead)
    og = to_oracle(orc, rmat12)
    for s in sources(rmat12, 3, seed=8):
        nsel, ids, vals, est, st = dev_rmat12.fora_topk(s, 0.5, ALPHA, 32, seed=6, cap=256, fetch=True)
        ref, sto = og.fora_topk(s, 0.5, ALPHA, 32, seed=6, schedule=orc.SYNC)
        assert st.rounds == sto.rounds
        assert st.walks == sto.walks
        assert st.levels == sto.levels and st.pops == sto.pops and st.dead_end_pops == sto.dead_end_pops
        assert_close(est, ref, TOL_MC, "topk est src=%d" % s)
        cnt, oids, ovals = orc.topk(ref, 32, cap=256)
        assert nsel == cnt and list(ids) == list(oids)


def test_topk_select_matches_oracle(pkg, orc, rmat12, dev_rmat12):
    og = to_oracle(orc, rmat12)
    s = sources(rmat12, 1, seed=33)[0]
    p, r, rsum, st = dev_rmat12.forward_push(s, ALPHA, 1e-6)
    for k in (1, 7, 32, 1000, rmat12.n + 5):
        nsel, ids, vals, kth, st = dev_rmat12.topk_select(k, cap=rmat12.n)
        cnt, oids, ovals = orc.topk(p, k)
        assert nsel == cnt
        assert list(ids) == list(oids)
        assert np.array_equal(vals, ovals)
        okth = orc.kth_largest(p, k)
        assert (okth is None and kth == 0.0) or okth == kth


# ------------------------------------------------------------------ backward search (a8, a9)
@pytest.mark.parametrize("rmax", [1e-3, 5e-5, 5e-7])
def test_backward_push_got(pkg, orc, got, dev_got, rmax):
    og = to_oracle(orc, got)
    for t in [0, 3, 17, 42, 106] + sources(got, 4, seed=13):
        p, r, st = dev_got.backward_push(t, ALPHA, rmax)
        po, ro, sto = og.backward_push(t, ALPHA, rmax, orc.SYNC)
        assert_close(p, po, TOL_PUSH, "bwd reserve t=%d" % t)
        assert_close(r, ro, TOL_PUSH, "bwd residue t=%d" % t)
        # dense levels count their nodes apart (the twin's backward search has one level shape)
        assert st.levels == sto.levels and st.pops + st.dense_nodes == sto.pops
        assert st.dense_levels > 0 or st.edge_pushes == sto.edge_pushes


def test_backward_push_rmat12(pkg, orc, rmat12, dev_rmat12):
    og = to_oracle(orc, rmat12)
    for t in sources(rmat12, 4, seed=17):
        p, r, st = dev_rmat12.backward_push(t, ALPHA, 1e-5)
        po, ro, sto = og.backward_push(t, ALPHA, 1e-5, orc.SYNC)
        assert_close(p, po, TOL_PUSH, "bwd reserve t=%d" % t)
        assert_close(r, ro, TOL_PUSH, "bwd residue t=%d" % t)


@pytest.mark.parametrize("dense_frac", [0.01, 0.05, 1e9])
def test_backward_push_level_shapes(pkg, orc, rmat12, dev_rmat12, rmat15, dev_rmat15, dense_frac):
    """Backward levels that touch much of the graph run as pull sweeps over the out-CSR; every switch point
    gives the twin's vectors."""
    t = pkg.tuning_default()
    t.dense_frac = dense_frac
    for host, dev, rmax in ((rmat12, dev_rmat12, 1e-7), (rmat15, dev_rmat15, 1e-6)):
        og = to_oracle(orc, host)
        dev.set_tuning(t)
        try:
            hub = int(np.argmax(np.diff(host.in_rp)))
            for tgt in [hub] + sources(host, 3, seed=12):
                p, r, st = dev.backward_push(tgt, ALPHA, rmax)
                po, ro, sto = og.backward_push(tgt, ALPHA, rmax, orc.SYNC)
                assert st.levels == sto.levels
                assert_close(p, po, TOL_PUSH, "backward reserve t=%d" % tgt)
                assert_close(r, ro, TOL_PUSH, "backward residue t=%d" % tgt)
            if dense_frac < 1:
                p, r, st = dev.backward_push(hub, ALPHA, rmax)
                assert st.dense_levels > 0          # the hub's search does go dense
        finally:
            dev.set_tuning(pkg.tuning_default())


@pytest.mark.parametrize("k", [-1, 3, 10])
def test_all_pair_backward_got(pkg, orc, got, dev_got, k):
    og = to_oracle(orc, got)
    ix, st = dev_got.all_pair_backward(ALPHA, 1e-3, k)
    off, tg, vl = ix.arrays()
    ooff, otg, ovl = og.all_pair_backward(ALPHA, 1e-3, k, schedule=orc.SYNC)
    assert np.array_equal(off, ooff)
    assert np.array_equal(tg, otg)
    assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
    ix.close()


@pytest.mark.parametrize("k", [-1, 0, 1, 2, 5, 9, 40])
def test_all_pair_k_rule_with_ties(pkg, orc, k):
    """The k rule of Base_Whole_Graph.java:112-163 where it is decided by ties: a source s with an edge to each of
    eight leaves (eight equal entries in its row), two parallel edges to h (a larger entry) and one to a chain
    (smaller ones).  Entries equal to the k-th largest all stay, in target order behind the larger ones; k > entries and
    k = 0 keep everything (value descending), k < 0 keeps target order.  The rows are ordered and cut on the device
    (kernels_sort.hip: three stable radix sorts, the rule as a prefix of each row); the oracle does it with a stable
    host sort."""
    src, dst = [], []
    s_, h_, c0 = 0, 1, 2                      # s, h, a chain c0 -> c1 -> c2
    leaves = list(range(5, 13))
    for t in leaves[::-1]:                    # (inserted in descending id order: the row order is not the edge order)
        src.append(s_); dst.append(t)
    src += [s_, s_, s_, c0, c0 + 1]
    dst += [h_, h_, c0, c0 + 1, c0 + 2]
    for t in leaves[:4]:                      # a second source with ties among four of the leaves
        src.append(13); dst.append(t)
    src += [13, 14]
    dst += [s_, 13]
    host = pkg.HostCsr(15, np.array(src, dtype=np.int32), np.array(dst, dtype=np.int32))
    og = to_oracle(orc, host)
    with pkg.Graph(host) as g:
        ix, _ = g.all_pair_backward(ALPHA, 1e-4, k)
        off, tg, vl = ix.arrays()
        ix.close()
    ooff, otg, ovl = og.all_pair_backward(ALPHA, 1e-4, k, schedule=orc.SYNC)
    assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
    assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
    row = vl[off[s_]:off[s_ + 1]]
    if k >= 0:
        assert np.all(np.diff(row) <= 0)                                   # value descending
    if k == 2:                                                             # s itself, h, and nothing of the eight ties
        assert len(row) == 2
    if k == 5:                                                             # the fifth largest is one of eight equal
        assert len(row) >= 3 + 8 and len(set(row[3:11].tolist())) == 1    # values: all eight stay, in target order
        assert tg[off[s_] + 3:off[s_] + 11].tolist() == leaves


@pytest.mark.parametrize("tier", ["1", "1-tables", "1-small", "2", "3"])
def test_all_pair_tiers_rmat12(pkg, orc, rmat12, dev_rmat12, tier, monkeypatch):
    """LDS hash tier, dense-vector tier and whole-vector (batch slot) tier give the same index (targets that outgrow
    a tier fall through to the next one on their own; the variable only moves the starting tier).  Tier 1 routes a
    target by its in-degree - small table, large table or straight on to the dense tier (kernels_apbs.hip:
    k_apbs_split): with the defaults, with every target sent through both tables, and with everything but the targets
    of the largest in-degrees started in the small one."""
    monkeypatch.setenv("PPRHIP_APBS_TIER", tier[0])
    if tier == "1-tables":
        monkeypatch.setenv("PPRHIP_APBS_DEG", "0,0")
    if tier == "1-small":
        monkeypatch.setenv("PPRHIP_APBS_DEG", "40,60")
    og = to_oracle(orc, rmat12)
    lo, hi = 100, 100 + (40 if tier == "3" else 300)
    for thr, k in ((1e-3, -1), (2e-4, 8)):
        ix, st = dev_rmat12.all_pair_backward(ALPHA, thr, k, lo, hi)
        off, tg, vl = ix.arrays()
        ooff, otg, ovl = og.all_pair_backward(ALPHA, thr, k, lo, hi, schedule=orc.SYNC)
        assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
        assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
        ix.close()


@pytest.mark.parametrize("cap_t,cap_f,after", [(8, 4096, "xl"), (4096, 3, "xl"), (4096, 3, "tier3"), (5, 2, "tier3"),
                                               (4096, 3, "whole"), (5, 2, "whole")])
def test_all_pair_dense_tier_list_overflows(pkg, orc, rmat12, cap_t, cap_f, after, monkeypatch):
    """The dense tier's lists are bounded: when the clean-up list overflows the search still finishes (the whole
    vector is cleared instead); when a frontier or the popped-node list overflows the search is run again - one at a
    time on the handle's own vectors ("whole", what a handful of such searches get by default), or on one of the few
    workspaces whose lists hold every node with the other workgroups helping ("xl": more of them than that), or - with
    both switched off - by the whole-vector tier on the batch slots ("tier3"); and the next search of the same workgroup
    must find all-zero vectors either way.  A fresh handle, so that the workspace is built with the shrunken lists."""
    monkeypatch.setenv("PPRHIP_APBS_TIER", "2")
    monkeypatch.setenv("PPRHIP_APBS_CAP_T", str(cap_t))
    monkeypatch.setenv("PPRHIP_APBS_CAP_F", str(cap_f))
    monkeypatch.setenv("PPRHIP_APBS_WHOLE", "1000000" if after == "whole" else "0")
    if after == "tier3":
        monkeypatch.setenv("PPRHIP_APBS_NO_XL", "1")
    og = to_oracle(orc, rmat12)
    with pkg.Graph(rmat12) as g:
        for lo, hi in ((100, 400), (0, 300)):            # twice: the second call starts from the vectors the first left
            ix, st = g.all_pair_backward(ALPHA, 5e-4, 6, lo, hi)
            off, tg, vl = ix.arrays()
            ooff, otg, ovl = og.all_pair_backward(ALPHA, 5e-4, 6, lo, hi, schedule=orc.SYNC)
            assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
            assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
            assert st.rounds == hi - lo                  # every target started in the dense tier
            if cap_f < 100:
                assert st.xl_targets > 0                 # searches outgrew the lists ...
                assert (st.dense_nodes > 0) == (after == "tier3")  # ... and reached the batch slots only with both passes off
            ix.close()
        # a single-target search on the same handle afterwards (the whole-vector pass used its vectors)
        p, r, _ = g.backward_push(7, ALPHA, 5e-4)
        po, ro, _ = og.backward_push(7, ALPHA, 5e-4, orc.SYNC)
        assert_close(p, po, TOL_PUSH, "backward search after All-Pair")


@pytest.mark.parametrize("hot,between", [(None, None), ("0", "0"), ("64", "1"), (None, "0")])
@pytest.mark.parametrize("chunk", [16, 256])
def test_all_pair_dense_tier_shared_levels(pkg, orc, rmat12, chunk, hot, between, monkeypatch):
    """Levels of the dense tier that span several chunks of edges are posted and idle workgroups take chunks of them
    (kernels_apbs.hip: work sharing).  With chunks of a few edges every level of every search is shared; the index
    must not depend on who pushed which edge.  hot: the ids whose residue and reserve live in the owner's LDS while a
    search is on its own (default: a quarter of this small graph; none; 64) - a posted level moves them into the global
    vector and back, so both sizes of level meet both kinds of id.  between: whether workgroups also help between two
    searches of their own (the default of a short pass) or only once they have run out of targets."""
    monkeypatch.setenv("PPRHIP_APBS_TIER", "2")
    monkeypatch.setenv("PPRHIP_APBS_CHUNK", str(chunk))
    if hot is not None:
        monkeypatch.setenv("PPRHIP_APBS_HOT", hot)
    if between is not None:
        monkeypatch.setenv("PPRHIP_APBS_HELP_BETWEEN", between)
    og = to_oracle(orc, rmat12)
    with pkg.Graph(rmat12) as g:
        for (lo, hi), thr in (((0, 64), 2e-4), ((1000, 1600), 1e-3)):   # few targets: most workgroups only help
            ix, st = g.all_pair_backward(ALPHA, thr, -1, lo, hi)
            off, tg, vl = ix.arrays()
            ooff, otg, ovl = og.all_pair_backward(ALPHA, thr, -1, lo, hi, schedule=orc.SYNC)
            assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
            assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
            ix.close()


def test_released_workspaces_come_back(pkg, rmat12):
    """pprhip_graph_release hands the All-Pair and batch workspaces back; the next call of those entry points
    allocates them again and returns what it returned before."""
    srcs = sources(rmat12, 5, seed=3)
    with pkg.Graph(rmat12) as g:
        g.set_tuning(pkg.tuning_batch())
        ix, _ = g.all_pair_backward(ALPHA, 1e-3, 8, 0, 512)
        a0 = [x.copy() for x in ix.arrays()]
        ix.close()
        b0 = g.fora_batch_single_source(srcs, 0.5, ALPHA, seed=9, k=8, fetch=True)
        g.release(g.RELEASE_ALL_PAIR | g.RELEASE_BATCH)
        g.release(g.RELEASE_ALL_PAIR | g.RELEASE_BATCH)  # nothing left to release: still fine
        ix, _ = g.all_pair_backward(ALPHA, 1e-3, 8, 0, 512)
        a1 = ix.arrays()
        assert all(np.array_equal(x, y) for x, y in zip(a0[:2], a1[:2])) and np.max(np.abs(a0[2] - a1[2])) <= TOL_PUSH
        ix.close()
        b1 = g.fora_batch_single_source(srcs, 0.5, ALPHA, seed=9, k=8, fetch=True)
        assert np.max(np.abs(b0[0] - b1[0])) <= 1e-9 and np.array_equal(b0[1], b1[1])
        with pytest.raises(pkg.PprhipError):
            g.release(64)


def test_batch_directions_share_a_handle(pkg, orc, rmat12, dev_rmat12, monkeypatch):
    """Forward batches and batched backward searches (All-Pair tier 3) alternate on one handle: the shared sweep
    arrays are handed over clean in both directions."""
    og = to_oracle(orc, rmat12)
    srcs = sources(rmat12, 9, seed=31)
    t = pkg.tuning_batch()
    dev_rmat12.set_tuning(t)
    monkeypatch.setenv("PPRHIP_APBS_TIER", "3")
    try:
        for _ in range(2):
            out, _, _, _, pq, _ = dev_rmat12.fora_batch_single_source(srcs, 0.5, ALPHA, seed=2, n_rounds=2, fetch=True,
                                                                      per_query=True)
            for i, s in enumerate(srcs):
                ref, sto = og.fora_whole(s, 0.5, ALPHA, seed=2, n_rounds=2, schedule=orc.SYNC,
                                         tuning=to_orc_tuning(orc, t))
                assert pq[i].walks == sto.walks
                assert_close(out[i], ref, TOL_MC, "forward batch src=%d" % s)
            ix, st = dev_rmat12.all_pair_backward(ALPHA, 2e-5, -1, 100, 140)
            off, tg, vl = ix.arrays()
            ooff, otg, ovl = og.all_pair_backward(ALPHA, 2e-5, -1, 100, 140, schedule=orc.SYNC)
            assert st.dense_levels > 0                      # the backward sweeps ran
            assert np.array_equal(off, ooff) and np.array_equal(tg, otg) and np.max(np.abs(vl - ovl)) <= 1e-12
            ix.close()
    finally:
        dev_rmat12.set_tuning(pkg.tuning_default())


def test_all_pair_whole_rmat12_counts(pkg, orc, rmat12, dev_rmat12):
    """Every target of the graph in one call; hub targets overflow the LDS table and are finished by the
    dense tier.  Pops and edge pushes equal the twin's."""
    og = to_oracle(orc, rmat12)
    ix, st = dev_rmat12.all_pair_backward(ALPHA, 1e-3, 4)
    off, tg, vl = ix.arrays()
    ooff, otg, ovl = og.all_pair_backward(ALPHA, 1e-3, 4, schedule=orc.SYNC)
    assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
    assert np.max(np.abs(vl - ovl)) <= TOL_PUSH
    ix.close()


def test_all_pair_sharded_merge(pkg, orc, got, dev_got):
    og = to_oracle(orc, got)
    a, _ = dev_got.all_pair_backward(ALPHA, 5e-4, 5, 0, 50)
    b, _ = dev_got.all_pair_backward(ALPHA, 5e-4, 5, 50, got.n)
    merged = pkg.merge_indexes([a, b], 5)
    off, tg, vl = merged.arrays()
    ooff, otg, ovl = og.all_pair_backward(ALPHA, 5e-4, 5, schedule=orc.SYNC)
    assert np.array_equal(off, ooff) and np.array_equal(tg, otg)
    assert np.max(np.abs(vl - ovl)) <= TOL_PUSH


# ------------------------------------------------------------------ ground truth (a12) and pure MC
def test_power_method(pkg, orc, got, dev_got, rmat12, dev_rmat12):
    for host, dev in ((got, dev_got), (rmat12, dev_rmat12)):
        og = to_oracle(orc, host)
        for s in sources(host, 3, seed=19):
            for iters in (1, 2, 100):
                p, st = dev.power_method(s, ALPHA, iters)
                po = og.power_method(s, ALPHA, iters)
                assert_close(p, po, TOL_PUSH, "power src=%d iters=%d" % (s, iters))


def test_monte_carlo(pkg, orc, got, dev_got):
    og = to_oracle(orc, got)
    for s in [0, 17, 106]:
        p, st = dev_got.monte_carlo(s, 0.5, ALPHA, seed=2)
        po, sto = og.monte_carlo(s, 0.5, ALPHA, seed=2)
        assert st.walks == sto.walks
        assert_close(p, po, TOL_MC, "mc src=%d" % s)


# ------------------------------------------------------------------ edge cases
def test_dead_end_and_isolated_sources(pkg, orc, toy_graphs):
    host = toy_graphs["isolated_mix"]
    og = to_oracle(orc, host)
    with pkg.Graph(host) as g:
        for s in (3, 4, 5):  # 3: dead end with in-edges; 4, 5: isolated
            est, st = g.fora_single_source(s, 0.5, ALPHA, seed=1, n_rounds=0)
            assert est[s] == 1.0 and est.sum() == 1.0
            nsel, ids, vals, e2, st = g.fora_topk(s, 0.5, ALPHA, 2, seed=1, cap=8, fetch=True)
            assert nsel == 1 and ids[0] == s and vals[0] == 1.0
            p, r, st = g.backward_push(s, ALPHA, 1e-4)
            po, ro, sto = og.backward_push(s, ALPHA, 1e-4, orc.SYNC)
            assert_close(p, po, TOL_PUSH, "bwd t=%d" % s)
        with pytest.raises(pkg.PprhipError):
            g.forward_push(host.n, ALPHA, 1e-3)
        with pytest.raises(pkg.PprhipError):
            g.forward_push(-1, ALPHA, 1e-3)


# ------------------------------------------------------------------ device-resident result store
def test_result_store_keeps_every_query(pkg, orc, rmat12, dev_rmat12):
    """pprhip_fora_batch_single_source_resident: every query's vector stays retrievable after its slot was reused
    (getWholeGraphPPR of query i, Gen_Util.java:309), identical to what reserve_out delivers; sums on the device."""
    srcs = sources(rmat12, 40, seed=21)  # more queries than slots: slots are reused
    store = pkg.Results(dev_rmat12, 40)
    try:
        assert store.info() == (40, 0, rmat12.n)
        out, ids, vals, nsel, _, _ = dev_rmat12.fora_batch_single_source(srcs, 0.5, ALPHA, seed=3, k=8, fetch=True,
                                                                         keep=store)
        assert store.info() == (40, 40, rmat12.n)
        for i in range(40):
            v = store.fetch(i)
            assert np.array_equal(v, out[i])
            assert abs(store.sum(i) - out[i].sum()) < 1e-12
            m = min(int(nsel[i]), 8)
            assert np.array_equal(v[ids[i][:m]], vals[i][:m])
        # a second call with fewer queries: the store reports what it holds now
        dev_rmat12.fora_batch_single_source(srcs[:3], 0.5, ALPHA, seed=3, keep=store)
        assert store.info()[1] == 3
        with pytest.raises(pkg.PprhipError):
            store.fetch(3)
        with pytest.raises(pkg.PprhipError):  # more queries than the store holds
            dev_rmat12.fora_batch_single_source(srcs + srcs, 0.5, ALPHA, seed=3, keep=store)
    finally:
        store.close()


# ------------------------------------------------------------------ Gauss-Seidel sweeps
def test_gauss_seidel_sweeps(pkg, orc, rmat15, dev_rmat15):
    """Dense levels cut into blocks (gs_blocks = 2, 3) against plain Jacobi sweeps (gs_blocks = 1): each schedule equals
    its twin level for level, every one of them ends in a state that meets the threshold and conserves mass, the
    blocked schedules need fewer dense levels, and FORA on top of them keeps its bound against the CPU power method."""
    og = to_oracle(orc, rmat15)
    od = np.diff(rmat15.out_rp).astype(np.float64)
    live = od > 0
    srcs = [s for s in sources(rmat15, 12, seed=31) if od[s] > 0][:3]
    dense = {}
    try:
        for B in (1, 2, 3):
            t = pkg.tuning_batch()
            t.gs_blocks = B
            dev_rmat15.set_tuning(t)   # (the twin's tuning follows: conftest)
            for s in srcs:
                p, r, rsum, st = dev_rmat15.forward_push(s, ALPHA, 1e-8)
                po, ro, _, sto = og.forward_push(s, ALPHA, 1e-8, orc.SYNC)
                assert_close(p, po, TOL_PUSH, "reserve B=%d src=%d" % (B, s))
                assert_close(r, ro, TOL_PUSH, "residue B=%d src=%d" % (B, s))
                assert st.levels == sto.levels and st.dense_levels == sto.dense_levels
                assert abs(p.sum() + r.sum() - 1.0) < 1e-12 and np.all(r[live] / od[live] < 1e-8)
                dense[(B, s)] = st.dense_levels
                est, stf = dev_rmat15.fora_single_source(s, 0.5, ALPHA, seed=3)
                ref, stfo = og.fora_whole(s, 0.5, ALPHA, seed=3, n_rounds=0, schedule=orc.SYNC, tuning=to_orc_tuning(orc, t))
                assert stf.walks == stfo.walks and stf.levels == stfo.levels
                assert_close(est, ref, TOL_MC, "FORA B=%d src=%d" % (B, s))
                pm = og.power_method(s, ALPHA, 100)
                big = pm > 1.0 / rmat15.n
                assert np.all(np.abs(est[big] - pm[big]) <= 0.5 * pm[big])
        for s in srcs:
            assert dense[(2, s)] < dense[(1, s)] and dense[(3, s)] < dense[(1, s)]
    finally:
        dev_rmat15.set_tuning(pkg.tuning_default())


# ------------------------------------------------------------------ round-2 layouts: slices, LDS table, trimmed rows
def test_sliced_sweep_layout(pkg, orc, rmat15, monkeypatch):
    """The single-query sweep over the sliced copy of the in-CSR (slices of 1 000 source ids here, so the R-MAT 15 has
    16 of them; the default width only slices graphs beyond 393 216 sources): same levels and values as the twin for
    Jacobi and block Gauss-Seidel sweeps, the same as the row-major layout, and the power method's sweeps too."""
    og = to_oracle(orc, rmat15)
    od = np.diff(rmat15.out_rp)
    srcs = [s for s in sources(rmat15, 12, seed=41) if od[s] > 0][:3]
    monkeypatch.setenv("PPRHIP_SLICE_IDS", "1000")
    g_sl = pkg.Graph(rmat15)
    monkeypatch.setenv("PPRHIP_SLICED", "0")
    g_rm = pkg.Graph(rmat15)
    try:
        for B in (1, 2, 3):
            t = pkg.tuning_batch()
            t.gs_blocks = B
            g_sl.set_tuning(t)
            g_rm.set_tuning(t)
            for s in srcs:
                p, r, rsum, st = g_sl.forward_push(s, ALPHA, 1e-8)
                po, ro, _, sto = og.forward_push(s, ALPHA, 1e-8, orc.SYNC)
                assert st.dense_levels > 0 and st.levels == sto.levels and st.dense_levels == sto.dense_levels
                assert_close(p, po, TOL_PUSH, "sliced reserve B=%d src=%d" % (B, s))
                assert_close(r, ro, TOL_PUSH, "sliced residue B=%d src=%d" % (B, s))
                p2, r2, _, st2 = g_rm.forward_push(s, ALPHA, 1e-8)
                assert st2.levels == st.levels and np.max(np.abs(p - p2)) <= TOL_PUSH
        pm, _ = g_sl.power_method(srcs[0], ALPHA, 30)
        assert_close(pm, og.power_method(srcs[0], ALPHA, 30), 1e-12, "power method over slices")
    finally:
        g_sl.set_tuning(pkg.tuning_default())
        g_sl.close()
        g_rm.close()


def test_sparse_push_lds_table(pkg, orc, rmat15, dev_rmat15, monkeypatch):
    """Sparse levels that sum a tile's contributions per destination in LDS before they land (forced on every level
    here; by default only levels of 2^18 edges and more): levels, pops and values as without the table and as the twin."""
    og = to_oracle(orc, rmat15)
    od = np.diff(rmat15.out_rp)
    srcs = [s for s in sources(rmat15, 12, seed=43) if od[s] > 0][:3]
    t = pkg.tuning_default()
    t.dense_frac = 4.0          # every level sparse: the largest ones carry hundreds of thousands of edges
    dev_rmat15.set_tuning(t)
    try:
        for s in srcs:
            monkeypatch.setenv("PPRHIP_COMB_MIN_EDGES", "1")
            p, r, rsum, st = dev_rmat15.forward_push(s, ALPHA, 1e-7)
            monkeypatch.setenv("PPRHIP_COMB_MIN_EDGES", "1000000000000")
            p2, r2, _, st2 = dev_rmat15.forward_push(s, ALPHA, 1e-7)
            po, ro, _, sto = og.forward_push(s, ALPHA, 1e-7, orc.SYNC)
            assert st.dense_levels == 0 and st.levels == st2.levels == sto.levels and st.pops == st2.pops == sto.pops
            assert_close(p, po, TOL_PUSH, "reserve with the table src=%d" % s)
            assert_close(r, ro, TOL_PUSH, "residue with the table src=%d" % s)
            assert np.max(np.abs(p - p2)) <= TOL_PUSH and abs(p.sum() + r.sum() - 1.0) < 1e-12
        tb, _ = dev_rmat15.backward_push(srcs[0], ALPHA, 1e-6)[0:2]
        monkeypatch.setenv("PPRHIP_COMB_MIN_EDGES", "1")
        tb2, _ = dev_rmat15.backward_push(srcs[0], ALPHA, 1e-6)[0:2]
        assert np.max(np.abs(tb - tb2)) <= TOL_PUSH
    finally:
        dev_rmat15.set_tuning(pkg.tuning_default())


def test_sparse_levels_all_modes(pkg, orc, rmat15, dev_rmat15, got, dev_got):
    """Batches of sparse levels (prepare + push per level, eight levels per host round trip) in every push mode: same
    levels, same counters, same vectors as the twin - the forward push (sparse levels only, and mixed with sweeps), the
    resumable top-k push round by round, and the backward search; on GOT (levels of a handful of edges) and R-MAT 15
    (levels of 10^5 edges)."""
    for host, dev, rmaxes in ((got, dev_got, (7.554e-4, 1e-8)), (rmat15, dev_rmat15, (1e-5, 1e-7))):
        og = to_oracle(orc, host)
        od = np.diff(host.out_rp)
        srcs = [s for s in sources(host, 12, seed=44) if od[s] > 0][:3]
        for dense_frac in (4.0, 0.05):
            t = pkg.tuning_default()
            t.dense_frac = dense_frac
            dev.set_tuning(t)
            try:
                for s in srcs:
                    for rmax in rmaxes:
                        p, r, rsum, st = dev.forward_push(s, ALPHA, rmax)
                        po, ro, rsum_o, sto = og.forward_push(s, ALPHA, rmax, orc.SYNC)
                        assert_close(p, po, TOL_PUSH, "reserve src=%d rmax=%g" % (s, rmax))
                        assert_close(r, ro, TOL_PUSH, "residue src=%d rmax=%g" % (s, rmax))
                        assert st.levels == sto.levels and st.dense_levels == sto.dense_levels
                        assert st.pops == sto.pops and st.edge_pushes == sto.edge_pushes
                        assert st.enqueues == sto.enqueues and st.dead_end_pops == sto.dead_end_pops
                        assert abs(p.sum() + r.sum() - 1.0) < 1e-12
                    # the resumable push of Fora_Topk's rounds (Forward_Push.java:144-250), thresholds falling by 4
                    dev.topk_push_reset(s, ALPHA)
                    tw = og.topk_push(s, ALPHA, orc.SYNC)
                    min_rmax = rmaxes[1]
                    for rnd in range(4):
                        rm = rmaxes[0] / 4.0 ** rnd
                        rs, st = dev.topk_push_round(min_rmax, rm)
                        rs_o, sto = tw.round(min_rmax, rm)
                        assert abs(rs - rs_o) <= 1e-12 and st.levels == sto.levels and st.pops == sto.pops
                        assert_close(dev.reserve(), tw.reserve, TOL_PUSH, "top-k round %d reserve src=%d" % (rnd, s))
                        assert_close(dev.residue(), tw.residue, TOL_PUSH, "top-k round %d residue src=%d" % (rnd, s))
                    pb, rb, stb = dev.backward_push(s, ALPHA, rmaxes[0])
                    pbo, rbo, stbo = og.backward_push(s, ALPHA, rmaxes[0], orc.SYNC)
                    assert_close(pb, pbo, TOL_PUSH, "backward reserve target=%d" % s)
                    assert stb.levels == stbo.levels and stb.pops + stb.dense_nodes == stbo.pops
            finally:
                dev.set_tuning(pkg.tuning_default())


def test_batched_sweep_sources_without_in_edges(pkg, orc, rmat15, dev_rmat15):
    """The batched sweep carries the rows with in-edges plus the rows without in-edges that have out-edges; a source of
    the second kind holds its own contribution (and the dead-end mass that returns to it) in a row nothing else ever
    writes.  Such sources next to ordinary ones: every query equals the single-query entry point and the twin."""
    og = to_oracle(orc, rmat15)
    od, idg = np.diff(rmat15.out_rp), np.diff(rmat15.in_rp)
    zin = np.nonzero((idg == 0) & (od > 0))[0]
    nz = np.nonzero((idg > 0) & (od > 0))[0]
    assert zin.size >= 12
    srcs = [int(x) for x in zin[:12]] + [int(x) for x in nz[:8]]
    t = pkg.tuning_batch()
    dev_rmat15.set_tuning(t)
    try:
        out, _, _, _, pq, st = dev_rmat15.fora_batch_single_source(srcs, 0.5, ALPHA, seed=7, fetch=True, per_query=True)
        assert st.class_launches[5] > 0
        for i, s in enumerate(srcs):
            assert abs(out[i].sum() - 1.0) < 1e-9
            single, sts = dev_rmat15.fora_single_source(s, 0.5, ALPHA, seed=7)
            assert sts.walks == pq[i].walks and sts.levels == pq[i].levels
            assert np.max(np.abs(single - out[i])) < 1e-9
        for i in (0, 5, 11, 14):
            ref, sto = og.fora_whole(srcs[i], 0.5, ALPHA, seed=7, n_rounds=0, schedule=orc.SYNC, tuning=to_orc_tuning(orc, t))
            assert sto.walks == pq[i].walks
            assert_close(out[i], ref, TOL_MC, "batched FORA, source without in-edges" if i < 12 else "batched FORA")
    finally:
        dev_rmat15.set_tuning(pkg.tuning_default())


def test_internal_order_is_invisible(pkg, orc, rmat15, monkeypatch):
    """PPRHIP_RELABEL=0 keeps the caller's vertex ids as the internal order (no hot-first layout, rows with in-edges
    scattered); with the sweeps' blocks off (they follow the internal order) both orders run the same levels and agree
    to rounding on every entry point - the internal order is a layout, not a semantic.  Sliced copy included."""
    od = np.diff(rmat15.out_rp)
    srcs = [s for s in sources(rmat15, 12, seed=47) if od[s] > 0][:2]
    monkeypatch.setenv("PPRHIP_SLICE_IDS", "3000")
    g_def = pkg.Graph(rmat15)
    monkeypatch.setenv("PPRHIP_RELABEL", "0")
    g_ids = pkg.Graph(rmat15)
    t = pkg.tuning_default()
    t.gs_blocks = 1
    try:
        for g in (g_def, g_ids):
            g.set_tuning(t)
        for s in srcs:
            a, b = g_def.forward_push(s, ALPHA, 1e-8), g_ids.forward_push(s, ALPHA, 1e-8)
            assert a[3].levels == b[3].levels and a[3].pops == b[3].pops and a[3].dense_levels == b[3].dense_levels > 0
            assert np.max(np.abs(a[0] - b[0])) <= TOL_PUSH and np.max(np.abs(a[1] - b[1])) <= TOL_PUSH
            ea, sa = g_def.fora_single_source(s, 0.5, ALPHA, seed=9)
            eb, sb = g_ids.fora_single_source(s, 0.5, ALPHA, seed=9)
            assert sa.walks == sb.walks and sa.walk_steps == sb.walk_steps and np.max(np.abs(ea - eb)) <= TOL_MC
            ba, bb = g_def.backward_push(s, ALPHA, 1e-6), g_ids.backward_push(s, ALPHA, 1e-6)
            assert np.max(np.abs(ba[0] - bb[0])) <= TOL_PUSH
            na, ia, va, _, _ = g_def.fora_topk(s, 0.5, ALPHA, 16, seed=4)
            nb, ib, vb, _, _ = g_ids.fora_topk(s, 0.5, ALPHA, 16, seed=4)
            assert na == nb and list(ia) == list(ib) and np.max(np.abs(va - vb)) <= TOL_MC
        xa, _ = g_def.all_pair_backward(ALPHA, 1e-3, 8, 100, 400)
        xb, _ = g_ids.all_pair_backward(ALPHA, 1e-3, 8, 100, 400)
        (oa, ta, va), (ob, tb, vb) = xa.arrays(), xb.arrays()
        assert np.array_equal(oa, ob) and np.array_equal(ta, tb) and np.max(np.abs(va - vb)) <= TOL_PUSH
        xa.close()
        xb.close()
    finally:
        g_def.close()
        g_ids.close()


@pytest.mark.gpu
def test_query_stream_equals_batched_calls(pkg, orc, rmat15, dev_rmat15):
    """pprhip_fora_stream_*: three submissions of ragged sizes run through one driver without a drain between them.
    Every query equals what the synchronous batched call gives for the same seed (and through it the twin): top-k
    blocks identical, stored vectors identical; a sample is held to the twin directly.  While the stream is open the
    handle's other entry points refuse; after the close they work again."""
    og = to_oracle(orc, rmat15)
    sizes, seeds = [5, 21, 1, 19], [5, 6, 7, 8]
    srcs = sources(rmat15, sum(sizes), seed=21)
    blocks = np.split(srcs, np.cumsum(sizes)[:-1])
    t = pkg.tuning_batch()
    dev_rmat15.set_tuning(t)
    store = pkg.Results(dev_rmat15, sum(sizes))
    ref_store = pkg.Results(dev_rmat15, max(sizes))
    try:
        with pkg.QueryStream(dev_rmat15, 0.5, ALPHA, k=8) as qs:
            tickets, first = [], 0
            for b, sd in zip(blocks, seeds):
                tickets.append(qs.submit(b, sd, keep=store, keep_first=first))
                first += b.size
            with pytest.raises(pkg.PprhipError) as ei:  # the driver thread owns the handle
                dev_rmat15.fora_single_source(int(srcs[0]), 0.5, ALPHA, seed=1)
            assert ei.value.code == pkg.ERR_STATE and "stream" in str(ei.value)
            with pytest.raises(pkg.PprhipError):
                qs.submit(np.array([rmat15.n], dtype=np.int32), 1)  # a source outside the graph: refused at submit
            with pytest.raises(pkg.PprhipError):
                qs.wait(999)
            got = [qs.wait(tk) for tk in reversed(tickets)][::-1]  # waits in any order
        assert store.info()[1] == sum(sizes)
        first = 0
        for b, sd, (ids, vals, nsel, st) in zip(blocks, seeds, got):
            _, ids2, vals2, nsel2, _, st2 = dev_rmat15.fora_batch_single_source(b, 0.5, ALPHA, seed=sd, k=8, keep=ref_store)
            # (walk deposits are fp64 atomics: equal up to the order of their sums)
            assert np.array_equal(nsel, nsel2) and np.array_equal(ids, ids2) and np.max(np.abs(vals - vals2)) <= 1e-12
            assert st.walks == st2.walks and st.levels == st2.levels and st.total_ms > 0
            for i in range(b.size):
                assert np.max(np.abs(store.fetch(first + i) - ref_store.fetch(i))) <= 1e-12
            ref, sto = og.fora_whole(b[0], 0.5, ALPHA, seed=sd, n_rounds=0, schedule=orc.SYNC, tuning=to_orc_tuning(orc, t))
            assert_close(store.fetch(first), ref, TOL_MC, "stream block seed %d src=%d" % (sd, b[0]))
            first += b.size
        # an empty stream opens and closes; a second stream on the same handle works
        pkg.QueryStream(dev_rmat15, 0.5, ALPHA, k=0).close()
        with pkg.QueryStream(dev_rmat15, 0.5, ALPHA, k=4) as qs:
            ids, vals, nsel, _ = qs.wait(qs.submit(blocks[0], 5))
        assert np.array_equal(ids[:, 0], got[0][0][:, 0])
    finally:
        store.close()
        ref_store.close()
        dev_rmat15.set_tuning(pkg.tuning_default())
