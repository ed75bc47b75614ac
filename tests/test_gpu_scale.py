"""GPU tests at benchmark scale (R-MAT 20, the configs[1] size; the engine's code paths are the same
at scale 22, which bench.py runs): size-independent properties of the path, since the CPU oracle
needs minutes per query here — mass conservation, the push invariant's sign and threshold
conditions, determinism under a fixed seed, top-k ordering, linearity of the push in the seed
mass (power method), agreement of the two vertex orders, and a sampled comparison with the oracle."""
import numpy as np
import pytest

from conftest import shared_graph, to_oracle

pytestmark = pytest.mark.gpu
A = 0.15


@pytest.fixture(scope="module")
def rmat20(pkg_product):
    pkg = pkg_product
    return pkg.HostCsr.rmat(20, 16, seed=1)


@pytest.fixture
def dev20(pkg, rmat20, dev_cache):
    return shared_graph(dev_cache, pkg, "dev20", lambda: pkg.Graph(rmat20))


def live_sources(host, count, seed):
    od = np.diff(host.out_rp)
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        s = int(rng.integers(0, host.n))
        if od[s] > 0:
            out.append(s)
    return out


def test_forward_push_conserves_mass_and_meets_threshold(rmat20, dev20):
    od = np.diff(rmat20.out_rp).astype(np.float64)
    for s in live_sources(rmat20, 3, 1):
        for rmax in (1e-6, 2.122e-8):
            p, r, rsum, st = dev20.forward_push(s, A, rmax)
            assert abs(p.sum() + r.sum() - 1.0) < 1e-11 and abs(rsum - r.sum()) < 1e-12
            assert p.min() >= 0.0 and r.min() >= 0.0
            live = od > 0
            assert np.all(r[live] / od[live] < rmax) and np.all(r[~live] == 0.0)  # Forward_Push.java:132
            assert st.levels > 0 and st.pops + st.dense_nodes >= st.levels


def test_fora_single_source_properties(orc, rmat20, dev20):
    og = to_oracle(orc, rmat20)
    for i, s in enumerate(live_sources(rmat20, 3, 2)):
        est, st = dev20.fora_single_source(s, 0.5, A, seed=5, n_rounds=0)
        assert abs(est.sum() - 1.0) < 1e-9 and est.min() >= 0.0
        assert st.walks >= int(st.omega * st.rsum) and st.rounds >= 1
        est2, st2 = dev20.fora_single_source(s, 0.5, A, seed=5, n_rounds=0)
        assert st2.walks == st.walks and st2.walk_steps == st.walk_steps
        assert np.max(np.abs(est - est2)) < 1e-12           # same seed, same walks; only the add order varies
        est3, _ = dev20.fora_single_source(s, 0.5, A, seed=6, n_rounds=0)
        assert np.max(np.abs(est - est3)) > 0                # another seed draws other walks
        # the estimate stays within FORA's relative bound of the 100-sweep ground truth where pi > delta; the ground
        # truth of the first source is the CPU's (Power_Method.java:44-101 in the oracle), which the GPU's must equal
        exact, _ = dev20.power_method(s, A, 100)
        if i == 0:
            cpu = og.power_method(s, A, 100)
            assert np.max(np.abs(exact - cpu)) <= 1e-12
            exact = cpu
        big = exact > 1.0 / rmat20.n
        assert np.mean(np.abs(est[big] - exact[big]) <= 0.5 * exact[big]) > 0.999


def test_fora_batch_at_scale(pkg, rmat20, dev20):
    """The batched entry point at benchmark scale: every query conserves mass, equals the single-query entry point
    (same thresholds, same walks; sums in another order), repeats exactly, and its top-32 follows the rule."""
    srcs = live_sources(rmat20, 20, 7) + [int(np.argmax(np.diff(rmat20.out_rp) == 0))]
    t = pkg.tuning_batch()
    dev20.set_tuning(t)
    try:
        out, ids, vals, nsel, pq, st = dev20.fora_batch_single_source(srcs, 0.5, A, seed=5, k=32, fetch=True,
                                                                      per_query=True)
        assert st.class_launches[5] > 0 and st.class_launches[5] < st.dense_levels    # sweeps shared by several queries
        out2, _, _, _, pq2, _ = dev20.fora_batch_single_source(srcs, 0.5, A, seed=5, fetch=True, per_query=True)
        assert np.max(np.abs(out - out2)) < 1e-12
        for i, s in enumerate(srcs):
            assert abs(out[i].sum() - 1.0) < 1e-9 and out[i].min() >= 0.0
            assert pq[i].walks == pq2[i].walks and pq[i].levels == pq2[i].levels
            m = min(int(nsel[i]), 32)
            assert np.all(np.diff(vals[i][:m]) <= 0) and np.array_equal(out[i][ids[i][:m]], vals[i][:m])
            if m == 32:
                assert int((out[i] >= vals[i][31]).sum()) == int(nsel[i])
        for i in (0, 9, 20):
            single, sts = dev20.fora_single_source(srcs[i], 0.5, A, seed=5, n_rounds=0)
            assert sts.walks == pq[i].walks and sts.rounds == pq[i].rounds and sts.levels == pq[i].levels
            assert np.max(np.abs(single - out[i])) < 1e-9
    finally:
        dev20.set_tuning(pkg.tuning_default())


def test_dead_end_sources_short_circuit(rmat20, dev20):
    od = np.diff(rmat20.out_rp)
    s = int(np.argmax(od == 0))
    est, st = dev20.fora_single_source(s, 0.5, A, seed=1)
    assert est[s] == 1.0 and est.sum() == 1.0 and st.walks == 0 and st.levels == 0
    n, ids, vals, _, _ = dev20.fora_topk(s, 0.5, A, 32, seed=1)
    assert n == 1 and ids[0] == s and vals[0] == 1.0


def test_topk_ordering_and_consistency(rmat20, dev20):
    for s in live_sources(rmat20, 3, 3):
        n, ids, vals, est, st = dev20.fora_topk(s, 0.5, A, 32, seed=9, cap=64, fetch=True)
        assert n >= 32 or n == int((est > 0).sum())
        assert np.all(np.diff(vals) <= 0)                       # value descending
        ties = np.diff(vals) == 0
        assert np.all(np.diff(ids)[ties] > 0)                   # ties by id ascending
        assert np.array_equal(est[ids], vals)
        kth = vals[min(n, 32) - 1]
        assert int((est >= kth).sum()) == n                     # every entry >= the k-th value is reported
        n2, ids2, vals2, kth2, _ = dev20.topk_select(32, cap=64)
        assert n2 == n and list(ids2) == list(ids) and kth2 == kth


def test_power_method_linearity_and_mass(rmat20, dev20):
    s = live_sources(rmat20, 1, 4)[0]
    p100, _ = dev20.power_method(s, A, 100)
    p50, _ = dev20.power_method(s, A, 50)
    assert abs(p100.sum() - (1 - (1 - A) ** 100)) < 1e-10    # undelivered mass after k sweeps = (1-alpha)^k
    assert abs(p50.sum() - (1 - (1 - A) ** 50)) < 1e-10
    assert np.all(p100 >= p50 - 1e-15)


def test_backward_push_and_all_pair_sample(orc, rmat20, dev20):
    og = to_oracle(orc, rmat20)
    idg = np.diff(rmat20.in_rp)
    rng = np.random.default_rng(6)
    for t in [int(x) for x in rng.integers(0, rmat20.n, 3)]:
        p, r, st = dev20.backward_push(t, A, 1e-4)
        po, ro, sto = og.backward_push(t, A, 1e-4, orc.SYNC)
        assert np.max(np.abs(p - po)) <= 1e-12 and st.pops + st.dense_nodes == sto.pops
        assert np.all(r <= 1e-4) and (idg[t] > 0 or p[t] == 1.0)
    lo = 5000
    ix, st = dev20.all_pair_backward(A, 1e-3, 8, lo, lo + 2000)
    off, tg, vl = ix.arrays()
    assert np.all(vl >= 1e-3) and np.all((tg >= lo) & (tg < lo + 2000))
    assert np.all(np.diff(off) <= np.maximum(8, np.diff(off)))   # k rule keeps >= k only on ties
    for v in np.nonzero(np.diff(off))[0][:200]:
        assert np.all(np.diff(vl[off[v]:off[v + 1]]) <= 0)
    ooff, otg, ovl = og.all_pair_backward(A, 1e-3, 8, lo, lo + 64, schedule=orc.SYNC)
    ix2, _ = dev20.all_pair_backward(A, 1e-3, 8, lo, lo + 64)
    off2, tg2, vl2 = ix2.arrays()
    assert np.array_equal(off2, ooff) and np.array_equal(tg2, otg) and np.max(np.abs(vl2 - ovl)) <= 1e-12


def column_sample(host, n_hubs, n_ranked, n_random, seed, t_lo=0, t_hi=None):
    """Targets for a column check: the n_hubs with the most in-edges (the searches whose levels are shared between
    workgroups and, where the lists overflow, the full-size pass), n_ranked more spread geometrically over the
    in-degree ranks behind them (the dense tier's middle), n_random drawn uniformly (mostly LDS-tier searches)."""
    t_hi = host.n if t_hi is None else t_hi
    ind = np.diff(host.in_rp)[t_lo:t_hi].astype(np.int64)
    order = np.argsort(-ind, kind="stable") + t_lo
    ranks = np.unique(np.geomspace(n_hubs + 1, min(order.size - 1, 200000), n_ranked).astype(np.int64))
    rng = np.random.default_rng(seed)
    return [int(x) for x in order[:n_hubs]] + [int(order[r]) for r in ranks] + \
           [int(x) for x in rng.integers(t_lo, t_hi, n_random)]


def check_columns_against_oracle(orc, og, arrays, targets, thr, k, fifo_for=None, workers=1):
    """Base_Whole_Graph.java:76-92 + the k rule (:112-163), column by column: the oracle's twin schedule to 1e-12 (same
    entries, same values), and the Java-faithful FIFO order under its bound (both orders leave every residue <= thr, so
    two reserves of one pair differ by at most thr: Backward_Search.java:89).  workers > 1: the oracle's searches run on
    that many host threads beforehand (ctypes calls release the GIL) and are kept as (ids, values) until they are
    compared - at R-MAT 24 a dense column is 134 MB and a hub's search takes the CPU seconds."""
    from bench import index_column_check
    off, tg, vl = arrays
    n = off.size - 1
    touched = {}
    fifo_for = targets if fifo_for is None else fifo_for

    def sparse_col(job):
        t, schedule = job
        p, r, _ = og.backward_push(t, A, thr, schedule)
        if schedule == orc.SYNC:
            touched[t] = int(((p > 0) | (r > 0)).sum())
        idx = np.nonzero(p)[0]
        return (t, schedule), (idx, p[idx])

    cache = {}
    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        jobs = [(int(t), orc.SYNC) for t in targets] + [(int(t), orc.FIFO) for t in fifo_for]
        with ThreadPoolExecutor(workers) as pool:
            cache = dict(pool.map(sparse_col, jobs))

    def column(t, schedule):
        key = (int(t), schedule)
        idx, val = cache.pop(key) if key in cache else sparse_col(key)[1]
        col = np.zeros(n)
        col[idx] = val
        return col

    st = index_column_check(off, tg, vl, targets, lambda t: column(t, orc.SYNC), thr, k, tol=1e-12)
    st_f = index_column_check(off, tg, vl, fifo_for, lambda t: column(t, orc.FIFO), thr, k, tol=1e-12, slack=thr)
    assert st_f["max_abs_diff"] <= thr
    return st, st_f, touched


@pytest.fixture(scope="module")
def full_index20(pkg_product, rmat20):
    pkg = pkg_product
    """All 2^20 targets of R-MAT 20 with the default settings: tier 1 in its three steps (targets routed by in-degree),
    the dense tier with levels shared at their natural sizes, the entries sorted on the device and the k rule applied
    by index_from_sorted."""
    with pkg.Graph(rmat20) as g:
        ix, st = g.all_pair_backward(A, 1e-3, 16)
        arrays = [x.copy() for x in ix.arrays()]
        ix.close()
    return arrays, st


@pytest.mark.timeout(900)
def test_all_pair_full_range_columns_against_oracle(orc, rmat20, full_index20):
    """The index of ALL targets, as the default path builds it, against oracle backward searches of 528
    targets: the 16 with the most in-edges, 112 across the dense tier's in-degree ranks, 400 at random."""
    og = to_oracle(orc, rmat20)
    arrays, st = full_index20
    targets = column_sample(rmat20, 16, 112, 400, seed=21)
    assert len(set(targets)) >= 512
    c, cf, touched = check_columns_against_oracle(orc, og, arrays, targets, 1e-3, 16)
    # the sample reaches both tiers (the LDS table gives up beyond 1536 nodes) and the k rule really cut entries
    sizes = np.array(list(touched.values()))
    assert (sizes <= 1536).sum() >= 100 and (sizes > 1536).sum() >= 100 and sizes.max() > 500000
    assert c["entries_checked"] > 100000 and c["entries_cut_by_k_rule"] > 0
    assert st.rounds > 0 and st.dense_nodes == 0          # dense-tier targets; nothing fell through to tier 3
    off, tg, vl = arrays
    assert np.all(vl >= 1e-3) and off[-1] == len(tg)


def test_all_pair_routing_does_not_change_the_index(pkg, rmat20, full_index20, monkeypatch):
    """Tier 1 routes every target by its in-degree - small LDS table, large one, or straight to the dense tier
    (kernels_apbs.hip: k_apbs_split).  The index of all 2^20 targets must be the one that sending every target through
    both tables gives (PPRHIP_APBS_DEG=0,0), and the dense tier must see more targets with the routing than without."""
    a, st = full_index20
    with pkg.Graph(rmat20) as g:
        monkeypatch.setenv("PPRHIP_APBS_DEG", "0,0")
        ix, st1 = g.all_pair_backward(A, 1e-3, 16)
        b = ix.arrays()
        # the same rows with the same entries; inside a row, entries whose values differ in the last bits (the
        # atomics' order) may swap places in the value order: compare by (row, target)
        assert np.array_equal(a[0], b[0])
        rows = np.repeat(np.arange(rmat20.n, dtype=np.int64), np.diff(a[0]).astype(np.int64))
        oa = np.argsort(rows * rmat20.n + a[1], kind="stable")
        ob = np.argsort(rows * rmat20.n + b[1], kind="stable")
        assert np.array_equal(a[1][oa], b[1][ob]) and np.max(np.abs(a[2][oa] - b[2][ob])) <= 1e-12
        for v in (a[2], b[2]):  # and every row is in value order
            inner = np.ones(len(v), dtype=bool)
            inner[a[0][1:-1][a[0][1:-1] < len(v)].astype(np.int64)] = False  # row starts
            assert np.all((np.diff(v) <= 0) | ~inner[1:])
        assert st.rounds >= st1.rounds > 0 and len(a[1]) > rmat20.n
        ix.close()


def test_sampled_walks_match_oracle(orc, rmat20, dev20):
    og = to_oracle(orc, rmat20)
    rng = np.random.default_rng(8)
    starts = rng.integers(0, rmat20.n, 2000).astype(np.int32)
    idx = rng.integers(0, 1 << 34, 2000).astype(np.uint64)
    term, steps = dev20.random_walks(starts, idx, A, seed=11, stream=2, no_zero_hop=True)
    for i in range(0, 2000, 7):
        assert og.random_walk(int(starts[i]), A, 11, 2, int(idx[i]), True) == (term[i], steps[i])


# ------------------------------------------------------------------ BASELINE.json's full sizes (configs #3-#5)
def _full_size_checks(pkg, scale, n_batch, more=None):
    """Size-independent properties at a benchmark size: the 36-bit packed (nodes | edges) counters, the uint32 edge
    offsets and the chunked sweep layout work on more than 2^26 edges; results conserve mass, meet the threshold,
    repeat, and the batched entry point equals the single-query one."""
    host = pkg.HostCsr.rmat(scale, 16, seed=1)
    od = np.diff(host.out_rp).astype(np.float64)
    live = od > 0
    srcs = live_sources(host, n_batch, 40 + scale)
    conf = pkg.conf_whole_graph(host.n, host.m, A)
    rmax0, omega = pkg.fora_whole_params(conf, 0.5)
    with pkg.Graph(host) as g:
        s = srcs[0]
        rmax = rmax0 / 32.0
        p, r, rsum, st = g.forward_push(s, A, rmax)
        assert abs(p.sum() + r.sum() - 1.0) < 1e-10 and abs(rsum - r.sum()) < 1e-11
        assert np.all(r[live] / od[live] < rmax) and np.all(r[~live] == 0.0) and p.min() >= 0.0
        assert st.dense_levels > 3 and host.m < st.dense_edges <= st.dense_levels * host.m  # several sweeps' worth
        est, st1 = g.fora_single_source(s, 0.5, A, seed=5)
        assert abs(est.sum() - 1.0) < 1e-9 and est.min() >= 0.0
        assert st1.walks >= int(st1.omega * st1.rsum) > 0
        n_sel, ids, vals, kth, _ = g.topk_select(32, cap=64)
        assert n_sel >= 32 and np.all(np.diff(vals) <= 0) and np.array_equal(est[ids], vals)
        assert int((est >= kth).sum()) == n_sel
        g.set_tuning(pkg.tuning_batch())
        store = pkg.Results(g, n_batch)
        try:
            _, bids, bvals, nsel, pq, stb = g.fora_batch_single_source(srcs, 0.5, A, seed=5, k=32, keep=store, per_query=True)
            assert stb.class_launches[5] > 0
            for i in range(n_batch):
                assert abs(store.sum(i) - 1.0) < 1e-9
                assert np.all(np.diff(bvals[i][:min(int(nsel[i]), 32)]) <= 0)
            v0 = store.fetch(0)
            est_b, st_b = g.fora_single_source(s, 0.5, A, seed=5)       # same profile, one at a time
            assert st_b.walks == pq[0].walks and st_b.levels == pq[0].levels
            assert np.max(np.abs(v0 - est_b)) < 1e-9
        finally:
            store.close()
            g.set_tuning(pkg.tuning_default())
        t0 = host.n // 3
        ix, sta = g.all_pair_backward(A, 1e-3, 32, t0, t0 + 2048)
        off, tg, vl = ix.arrays()
        assert np.all(vl >= 1e-3) and np.all((tg >= t0) & (tg < t0 + 2048)) and sta.pops >= 2048
        ix.close()
        if more is not None:
            more(host, g)


@pytest.mark.timeout(900)
def test_full_size_rmat22(pkg):
    _full_size_checks(pkg, 22, 6)


@pytest.mark.timeout(1500)
def test_full_size_rmat24(pkg, orc):
    """config #5's graph (n = 16.7 M, m = 268 M) - and config #5's own workload held to the oracle: the index of ALL
    16.7 M targets (threshold 1e-3, k = 32, default settings: Base_Whole_Graph.java:76-92 and the k rule of :112-163),
    32 of its columns against oracle backward searches (Backward_Search.java:38-100) - the 8 targets with the most
    in-edges, 8 across the in-degree ranks, 16 at random: the twin schedule to 1e-12 with exact membership both ways,
    the Java-faithful FIFO order under the bound the two orders share (:84-89), entries missing from a row accepted
    only where the k rule cut them."""
    def all_pair_all_targets(host, g):
        batch = g.RELEASE_BATCH
        g.release(batch)   # (the batched queries' workspaces are not needed beside All-Pair's 77 GB)
        ix, st = g.all_pair_backward(A, 1e-3, 32)
        arrays = [x.copy() for x in ix.arrays()]
        ix.close()
        g.release(g.RELEASE_ALL_PAIR)
        off, tg, vl = arrays
        assert off[-1] == len(tg) > host.n and np.all(vl >= 1e-3) and st.pops >= host.n
        assert st.dense_nodes == 0                          # nothing was left for tier 3
        og = to_oracle(orc, host)
        targets = column_sample(host, 8, 8, 16, seed=24)
        assert len(set(targets)) >= 30
        c, cf, touched = check_columns_against_oracle(orc, og, arrays, targets, 1e-3, 32, workers=8)
        sizes = np.array(list(touched.values()))
        print("R-MAT 24 All-Pair: %d entries, %d columns checked (%d entries, %d cut by the k rule), searches touch %d .. %d nodes,"
              " max |diff| twin %.2e, FIFO %.2e" % (len(tg), c["targets"], c["entries_checked"], c["entries_cut_by_k_rule"],
                                                   sizes.min(), sizes.max(), c["max_abs_diff"], cf["max_abs_diff"]))
        assert sizes.max() > (1 << 20) and (sizes <= 1536).sum() >= 4   # both ends of the tiers are in the sample
        assert c["entries_checked"] > 50000 and c["entries_cut_by_k_rule"] > 0

    _full_size_checks(pkg, 24, 3, more=all_pair_all_targets)


@pytest.mark.timeout(1200)
def test_full_size_rmat22_against_cpu_power_method(pkg, orc):
    """Benchmark size against the CPU ground truth (Power_Method.java:44-101, 100 sweeps, ~0.5 min of host time per
    source, the three in parallel) for three sources - the node with the most out-edges, a node with one out-edge and a
    random live one: the GPU power method equals it to 1e-12, FORA (single and batched entry points, eps = 0.5) keeps
    its relative bound on every node with pi > delta (Fora_Whole_Graph's guarantee; delta = 1/n), the top-32 agree
    wherever the exact values are further apart than the estimate's error, and Fora_Topk at k = 32 (config #3;
    Fora_Topk.java:102-184) through the single and the batched entry point keeps its eps / 2 bound."""
    from concurrent.futures import ThreadPoolExecutor
    host = pkg.HostCsr.rmat(22, 16, seed=1)
    og = to_oracle(orc, host)
    od = np.diff(host.out_rp)
    # (the degree-1 source: one whose only neighbour has out-edges of its own - through a dead end all mass comes back and
    # two nodes hold all of it)
    one = np.nonzero(od == 1)[0]
    one = one[od[host.out_ci[host.out_rp[one]]] >= 10]
    srcs = [int(np.argmax(od)), int(one[1234]), live_sources(host, 1, 77)[0]]
    assert od[srcs[0]] > 10000 and od[srcs[1]] == 1 and len(set(srcs)) == 3
    with ThreadPoolExecutor(3) as pool:   # (ctypes calls release the GIL)
        exacts = list(pool.map(lambda s: og.power_method(s, A, 100), srcs))
    k = 32
    with pkg.Graph(host) as g:
        g.set_tuning(pkg.tuning_batch())
        try:
            out, ids, vals, nsel, pq, _ = g.fora_batch_single_source(srcs, 0.5, A, seed=6, k=k, fetch=True, per_query=True)
        finally:
            g.set_tuning(pkg.tuning_default())
        bids, bvals, bst = g.fora_batch_topk(np.array(srcs + srcs, dtype=np.int32), k, 0.5, A, seed=5)
        for i, (s, exact) in enumerate(zip(srcs, exacts)):
            pm, _ = g.power_method(s, A, 100)
            assert np.max(np.abs(pm - exact)) <= 1e-12
            big = exact > 1.0 / host.n
            est, st = g.fora_single_source(s, 0.5, A, seed=5)
            for e, est_st in ((est, st), (out[i], pq[i])):
                # floor(omega * rsum) = 0 walks (the hub's push runs until next to nothing is left) leaves the residues
                # undelivered, as in the reference (Fora_Whole_Graph.java:112-119)
                assert est_st.walks > 0 or est_st.omega * est_st.rsum < 1.0
                assert abs(e.sum() + (est_st.rsum if est_st.walks == 0 else 0.0) - 1.0) < 1e-9
                err = np.abs(e - exact)
                assert np.all(err[big] <= 0.5 * exact[big])
                # top-32: same set up to swaps among values closer than twice the largest error seen on the top entries
                order = np.argsort(-exact, kind="stable")[:64]
                tol = 2.0 * err[order].max()
                top_exact, top_est = set(order[:32].tolist()), set(np.argsort(-e, kind="stable")[:32].tolist())
                kth = exact[order[31]]
                for v in top_exact ^ top_est:
                    assert abs(exact[v] - kth) <= tol
            m = min(int(nsel[i]), k)
            assert np.array_equal(out[i][ids[i][:m]], vals[i][:m])
            # ---- Fora_Topk.computeTopKPPR at full size: the single entry point and the batched one
            order = np.argsort(-exact, kind="stable")
            kth = exact[order[k - 1]]
            n_sel, tids, tvals, test_, tst = g.fora_topk(s, 0.5, A, k, seed=5 + i, cap=4 * k, fetch=True)
            assert tst.rounds >= 1 and tst.walks > 0 and n_sel >= k
            assert np.all(np.diff(tvals) <= 0) and np.array_equal(test_[tids], tvals)
            # query i of the batch runs with the same seed as the single call: same rounds, same walks, same list
            assert np.array_equal(bids[i], tids[:k]) and np.max(np.abs(bvals[i] - tvals[:k])) <= 1e-9
            for ids_k, vals_k in ((tids[:k], tvals[:k]), (bids[3 + i], bvals[3 + i])):
                # the stopping rule (:175) bounds the relative error of the reported entries by eps' = eps / 2
                assert np.all(np.abs(vals_k - exact[ids_k]) <= 0.25 * np.maximum(exact[ids_k], kth))
                # top-32 identity wherever the exact k-th and (k+1)-th values are further apart than that error (gap
                # guard); otherwise only entries within the error of the k-th place may differ
                gap = exact[order[k - 1]] - exact[order[k]]
                if gap > 2 * 0.25 * kth:
                    assert set(ids_k.tolist()) == set(order[:k].tolist())
                else:
                    for v in set(ids_k.tolist()) ^ set(order[:k].tolist()):
                        assert abs(exact[v] - kth) <= 2 * 0.25 * kth
            # the whole estimate the last round leaves (Fora_Topk.java:143-168): every residue's walks deliver all of it
            # (omega_v = ceil(r W) >= 1 walks of r / omega_v each, :155-167), so reserve + walk increments sum to 1
            assert abs(test_.sum() - 1.0) < 1e-9


@pytest.mark.timeout(1200)
def test_full_size_rmat22_all_pair_columns_against_oracle(pkg, orc):
    """All 4.19 M targets of R-MAT 22 (the bench's All-Pair workload: threshold 1e-3, k = 32) on one GPU with the
    default settings - targets routed by in-degree, levels shared at their natural sizes, a whole-vector search for the
    target whose lists outgrow a workspace, device sort + index_from_sorted - and 56 of its columns against oracle backward
    searches: the 8 targets with the most in-edges, 16 across the in-degree ranks, 32 at random
    (Base_Whole_Graph.java:76-92,112-163; Backward_Search.java:84-89)."""
    host = pkg.HostCsr.rmat(22, 16, seed=1)
    og = to_oracle(orc, host)
    with pkg.Graph(host) as g:
        ix, st = g.all_pair_backward(A, 1e-3, 32)
        arrays = ix.arrays()
        ix.close()
    assert st.xl_targets >= 1 and st.dense_nodes == 0     # a search outgrew the lists; nothing was left for tier 3
    targets = column_sample(host, 8, 16, 32, seed=22)
    c, cf, touched = check_columns_against_oracle(orc, og, arrays, targets, 1e-3, 32, fifo_for=targets[:4] + targets[24:40])
    assert max(touched.values()) > (1 << 20)              # a search beyond the per-workgroup lists was among them
    assert c["entries_checked"] > 100000 and c["entries_cut_by_k_rule"] > 0


@pytest.mark.timeout(1200)
def test_full_size_rmat22_fifty_query_call_and_stream(pkg, orc):
    """Config #4's call shape at its own size: ONE call of 50 sources (PPR.java:179's default; the loop of
    Gen_Util.java:208-232) through pprhip_fora_batch_single_source_resident - three full rounds of 16 slots and the two
    leftover queries that run singly - and the same 50 sources as one submission of a query stream.  Three of the 50
    (the first, leftover query 49, one from the middle) are held to the CPU power method (Power_Method.java:44-101) under
    FORA's (eps, delta) bound; the stream must equal the synchronous call: same top-32 lists, vectors to 1e-12."""
    from concurrent.futures import ThreadPoolExecutor
    host = pkg.HostCsr.rmat(22, 16, seed=1)
    og = to_oracle(orc, host)
    srcs = np.array(live_sources(host, 50, 404), dtype=np.int32)
    held = [0, 23, 49]
    with ThreadPoolExecutor(3) as pool:
        exact_f = [pool.submit(og.power_method, int(srcs[i]), A, 100) for i in held]
        with pkg.Graph(host) as g:
            g.set_tuning(pkg.tuning_batch())
            sync_store, stream_store = pkg.Results(g, 50), pkg.Results(g, 50)
            try:
                _, ids, vals, nsel, pq, st = g.fora_batch_single_source(srcs, 0.5, A, seed=9, k=32, keep=sync_store,
                                                                        per_query=True)
                assert st.class_launches[5] > 0 and all(p.walks > 0 or p.omega * p.rsum < 1.0 for p in pq)
                with pkg.QueryStream(g, 0.5, A, k=32) as qs:
                    ids_s, vals_s, nsel_s, st_s = qs.wait(qs.submit(srcs, 9, keep=stream_store))
                assert np.array_equal(nsel, nsel_s) and np.array_equal(ids, ids_s)
                assert np.max(np.abs(vals - vals_s)) <= 1e-12 and st_s.walks == st.walks
                for i in range(50):
                    s_sync, s_stream = sync_store.sum(i), stream_store.sum(i)
                    lost = pq[i].rsum if pq[i].walks == 0 else 0.0   # (floor(omega rsum) = 0 walks leave the residues)
                    assert abs(s_sync + lost - 1.0) < 1e-9 and abs(s_stream - s_sync) < 1e-11
                    m = min(int(nsel[i]), 32)
                    assert m == 32 and np.all(np.diff(vals[i][:m]) <= 0)
                for i, fut in zip(held, exact_f):
                    exact = fut.result()
                    v_sync, v_stream = sync_store.fetch(i), stream_store.fetch(i)
                    assert np.max(np.abs(v_sync - v_stream)) <= 1e-12
                    assert np.array_equal(v_sync[ids[i]], vals[i])
                    big = exact > 1.0 / host.n                      # Fora_Whole_Graph's guarantee: pi > delta = 1/n
                    err = np.abs(v_sync - exact)
                    assert np.all(err[big] <= 0.5 * exact[big]), "query %d (source %d) leaves FORA's bound" % (i, srcs[i])
                    order = np.argsort(-exact, kind="stable")[:64]
                    tol = 2.0 * err[order].max()
                    for v in set(order[:32].tolist()) ^ set(ids[i].tolist()):
                        assert abs(exact[v] - exact[order[31]]) <= tol
            finally:
                sync_store.close()
                stream_store.close()
                g.set_tuning(pkg.tuning_default())


@pytest.mark.timeout(900)
def test_full_size_rmat22_row_panel_sweep_equals_sliced_sweep(pkg, monkeypatch):
    """The single-query forward sweep's two layouts at BASELINE's size (R-MAT 22: the row-panel copy is the default from
    2^26 edges on, PPRHIP_SWEEP1_PANELS=0 forces the sliced copy): forward push, the power method and whole-graph FORA
    from a hub, a degree-1 source and a random live one - the same levels, the same walks, vectors equal up to the order
    of the sums (1e-12), mass conserved."""
    host = pkg.HostCsr.rmat(22, 16, seed=1)
    od = np.diff(host.out_rp)
    srcs = [int(np.argmax(od)), int(np.nonzero(od == 1)[0][0]), live_sources(host, 1, 4)[0]]
    results = {}
    for layout in ("1", "0"):
        monkeypatch.setenv("PPRHIP_SWEEP1_PANELS", layout)
        with pkg.Graph(host) as g:
            out = []
            for s in srcs:
                p, r, rsum, st = g.forward_push(s, A, 5.07e-9)
                assert st.dense_levels > 0 and abs(p.sum() + r.sum() - 1.0) < 1e-11
                est, stf = g.fora_single_source(s, 0.5, A, seed=3)
                assert abs(est.sum() - 1.0) < 1e-9
                out.append((p, r, st.levels, st.dense_levels, est, stf.walks, stf.levels))
            pm, _ = g.power_method(srcs[0], A, 20)
            out.append(pm)
        results[layout] = out
    for a, b in zip(results["1"][:-1], results["0"][:-1]):
        assert a[2] == b[2] and a[3] == b[3] and a[5] == b[5] and a[6] == b[6]
        assert np.max(np.abs(a[0] - b[0])) <= 1e-12 and np.max(np.abs(a[1] - b[1])) <= 1e-12
        assert np.max(np.abs(a[4] - b[4])) <= 1e-9
    assert np.max(np.abs(results["1"][-1] - results["0"][-1])) <= 1e-12
