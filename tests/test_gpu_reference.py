"""GPU parity against the Java-faithful side, independent of the frontier-synchronous twin.

tests/test_gpu_parity.py holds the HIP engine to its bit-level twin (`orc.SYNC`), which is written
by the same hand as the engine.  This file closes the chain on the GPU without it: the engine
(through the C ABI) is compared directly with

  * the reference's own schedule, `orc.FIFO` (Forward_Push.java:79-141, Backward_Search.java:51-97),
  * the CPU power method (Power_Method.java:44-101), the reference's ground truth,
  * the committed vectors of tests/golden/got_fifo_golden.json (frozen FIFO / power-method outputs
    on the reference's Game-of-Thrones dataset; generator tests/golden/make_fifo_golden.py),

at `north_star`'s bar: reserve vectors within 1e-6 L-inf, top-k sets identical (sources whose exact
k-th and (k+1)-th values are closer than the error bound are skipped: SURVEY.md §7 hard part 1 (iv)).
Two push schedules agree only up to what they leave in the residues, so deterministic comparisons
run at thresholds whose residue bound m * rmax (forward) or rmax (backward) is far below 1e-6; FORA
results are compared with the power method under FORA's own (eps, delta) guarantee
(Fora_Whole_Graph.java:86-87).
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT, edges_to_host, to_oracle

pytestmark = pytest.mark.gpu

A = 0.15
EPS = 0.5
TOL_SPEC = 1e-6  # north_star: "reserve vectors within 1e-6 L-inf"
GOLDEN = os.path.join(ROOT, "tests", "golden", "got_fifo_golden.json")


def unhex(v):
    return np.array([float.fromhex(x) for x in v])


@pytest.fixture(scope="module")
def golden():
    return json.load(open(GOLDEN))


@pytest.fixture(scope="module")
def dev_got(pkg_product, got):
    pkg = pkg_product
    g = pkg.Graph(got)
    yield g
    g.close()


@pytest.fixture(scope="module")
def dev_rmat12(pkg_product, rmat12):
    pkg = pkg_product
    g = pkg.Graph(rmat12)
    yield g
    g.close()


@pytest.fixture(scope="module")
def got_undirected(pkg_product, got):
    pkg = pkg_product
    """GOT with every relationship in both directions: no dead ends, the setting the thesis states for
    backward search (Diss. p.19, p.31), so that its columns equal the power method's."""
    e = []
    for v in range(got.n):
        for u in got.out_ci[got.out_rp[v]:got.out_rp[v + 1]]:
            e.append((v, int(u)))
            e.append((int(u), v))
    return edges_to_host(pkg, got.n, e)


def live_sources(host, count, seed):
    od = np.diff(host.out_rp)
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        s = int(rng.integers(0, host.n))
        if od[s] > 0:
            out.append(s)
    return out


def exact_ppr_matrix(host, s, alpha):
    """alpha (I - (1-alpha) P_s)^-1 where dead-end rows of P_s point to the query source s (Forward_Push.java:101-113):
    row v is the distribution of a walk started at v under the reference's dynamics for source s.  Plain numpy."""
    n = host.n
    P = np.zeros((n, n))
    for v in range(n):
        nb = host.out_ci[host.out_rp[v]:host.out_rp[v + 1]]
        if len(nb) == 0:
            P[v, s] = 1.0
        else:
            np.add.at(P[v], nb, 1.0 / len(nb))
    return alpha * np.linalg.inv(np.eye(n) - (1 - alpha) * P)


def topk_gap_ok(exact, k, tol):
    """At least k + 1 exact entries, and the k-th and (k+1)-th values are further apart than twice the tolerance
    (else set identity is not decidable at that tolerance; exact ties by symmetry fall here too)."""
    v = np.sort(exact[exact > 0])[::-1]
    return v.size > k and v[k - 1] - v[k] > 2 * tol


# ------------------------------------------------------------------ (e) committed FIFO vectors
def test_engine_against_committed_fifo_vectors(golden, got, dev_got):
    """The engine reproduces the frozen Java-faithful vectors: push at rmax = 1e-10 within 1e-6 (in fact
    within m * rmax = 3.5e-8) of the FIFO reserve and of the power method, exact top-10 sets."""
    assert golden["graph"] == {"n": got.n, "m": got.m, "dataset": golden["graph"]["dataset"]}
    for name, e in golden["sources"].items():
        s = e["id"]
        pm = unhex(e["power_method_100"])
        gp = e["forward_push_1e-10"]
        p, r, rsum, st = dev_got.forward_push(s, A, float.fromhex(gp["rmax"]))
        fifo = unhex(gp["reserve"])
        assert np.max(np.abs(p - fifo)) <= 2 * got.m * 1e-10 <= TOL_SPEC, name
        assert np.max(np.abs(p - pm)) <= TOL_SPEC, name
        assert abs(p.sum() + r.sum() - 1.0) < 1e-12 or e["out_degree"] == 0
        # the GPU's power method against the frozen CPU one (same Jacobi sweeps; only fp64 add order differs)
        gpm, _ = dev_got.power_method(s, A, 100)
        assert np.max(np.abs(gpm - pm)) <= 1e-12, name
        # top-10 set of the push result == frozen power-method top-10 (ties at the 10th value included by the rule)
        dev_got.forward_push(s, A, 1e-10, fetch=False)
        n_sel, ids, vals, kth, _ = dev_got.topk_select(10, cap=got.n)
        want = e["power_method_top10"]
        if topk_gap_ok(pm, 10, 4e-8):  # Tyrion's 10th and 11th exact values tie: not decidable, skipped
            assert n_sel == want["count"] == 10 and sorted(ids.tolist()) == sorted(want["ids"]), name
        elif want["count"] < 10:       # fewer than 10 entries (dead-end source): the whole map
            assert set(ids.tolist()) <= set(want["ids"])
        # backward search at rmax = 1e-8: both schedules leave residues <= rmax, so reserves agree to 2e-8
        bp = e["backward_push_1e-8"]
        q, qr, _ = dev_got.backward_push(s, A, 1e-8)
        assert np.max(np.abs(q - unhex(bp["reserve"]))) <= 2e-8 <= TOL_SPEC, name
        assert qr.max() <= 1e-8
        # the reference's own first threshold rmax0 (Fora_Whole_Graph.java:86): FIFO and the engine stop in different
        # states; each must be a push state of the SAME exact vector (dense solve, independent of oracle and engine):
        # pi_s = reserve + sum_v residue(v) pi_s(v, .), every residue below the threshold - so the two reserves differ
        # by exactly what the two residue vectors still hold
        g0 = e["forward_push_rmax0"]
        rmax0 = float.fromhex(g0["rmax"])
        p0, r0, _, _ = dev_got.forward_push(s, A, rmax0)
        pf, rf = unhex(g0["reserve"]), unhex(g0["residue"])
        od = np.diff(got.out_rp)
        if od[s] > 0:
            Pi = exact_ppr_matrix(got, s, A)
            assert np.max(np.abs(p0 + r0 @ Pi - Pi[s])) < 1e-12, name
            assert np.max(np.abs(pf + rf @ Pi - Pi[s])) < 1e-12, name
            assert np.max(np.abs((p0 - pf) - (rf - r0) @ Pi)) < 1e-12, name
            assert np.all((od == 0) & (r0 == 0) | (od > 0) & (r0 / np.maximum(od, 1) < rmax0)), name
        else:
            assert p0[s] == 1.0 and r0.sum() == 0.0 and np.array_equal(p0, pf), name


# ------------------------------------------------------------------ (a) push vs FIFO and vs the CPU power method
def test_forward_push_vs_fifo_and_power_method(pkg, orc, got, dev_got, toy_graphs, rmat12, dev_rmat12):
    cases = [(got, dev_got, 1e-10, list(range(0, got.n, 9)) + [63, 71, 1])]
    cases.append((rmat12, dev_rmat12, 1e-12, live_sources(rmat12, 3, 11) + [int(np.argmax(np.diff(rmat12.out_rp) == 0))]))
    for host, dev, rmax, srcs in cases:
        og = to_oracle(orc, host)
        assert host.m * rmax < 1e-7
        for s in srcs:
            p, r, rsum, st = dev.forward_push(s, A, rmax)
            pf, rf, _, _ = og.forward_push(s, A, rmax, orc.FIFO)
            pm = og.power_method(s, A, 100)
            assert np.max(np.abs(p - pf)) <= TOL_SPEC and np.max(np.abs(p - pf)) <= 2 * host.m * rmax
            assert np.max(np.abs(p - pm)) <= TOL_SPEC
            # top-k set identity vs the power method (k = 10 and 32), gap guard at the error bound
            dev.forward_push(s, A, rmax, fetch=False)
            for k in (10, 32):
                cnt, oids, _ = orc.topk(pm, k, cap=host.n)
                n_sel, ids, _, _, _ = dev.topk_select(k, cap=host.n)
                if topk_gap_ok(pm, k, host.m * rmax + 1e-7):
                    assert n_sel == cnt == k and set(ids.tolist()) == set(oids.tolist()), (s, k)
                elif cnt < k:  # fewer than k reachable nodes: every entry of the map is reported
                    assert set(ids.tolist()) <= set(oids.tolist())
    for name, host in toy_graphs.items():
        og = to_oracle(orc, host)
        with pkg.Graph(host) as g:
            for s in range(host.n):
                p, _, _, _ = g.forward_push(s, A, 1e-12)
                pf, _, _, _ = og.forward_push(s, A, 1e-12, orc.FIFO)
                assert np.max(np.abs(p - pf)) <= 1e-9, (name, s)
                assert np.max(np.abs(p - og.power_method(s, A, 100))) <= TOL_SPEC, (name, s)


def test_backward_push_vs_fifo_and_power_method_columns(pkg, orc, got, dev_got, got_undirected, rmat12, dev_rmat12):
    og = to_oracle(orc, got)
    for t in (63, 71, 1, 0, 50, 106):
        q, qr, _ = dev_got.backward_push(t, A, 1e-9)
        qf, _, _ = og.backward_push(t, A, 1e-9, orc.FIFO)
        assert np.max(np.abs(q - qf)) <= 2e-9 and qr.max() <= 1e-9
    og12 = to_oracle(orc, rmat12)
    for t in live_sources(rmat12, 3, 5):
        q, qr, _ = dev_rmat12.backward_push(t, A, 1e-8)
        qf, _, _ = og12.backward_push(t, A, 1e-8, orc.FIFO)
        assert np.max(np.abs(q - qf)) <= 2e-8 <= TOL_SPEC
    # without dead ends pi^b(v, t) is column t of the power method's matrix (Diss. p.30)
    ou = to_oracle(orc, got_undirected)
    cols = np.stack([ou.power_method(v, A, 100) for v in range(got_undirected.n)])  # cols[v, t] = pi(v, t)
    with pkg.Graph(got_undirected) as g:
        for t in (63, 0, 17, 99):
            q, _, _ = g.backward_push(t, A, 1e-9)
            assert np.max(np.abs(q - cols[:, t])) <= TOL_SPEC


# ------------------------------------------------------------------ (b) FORA vs the CPU power method
def fora_bound_ok(est, exact, eps, delta):
    big = exact > delta
    return np.all(np.abs(est[big] - exact[big]) <= eps * exact[big])


def test_fora_vs_cpu_power_method(pkg, orc, got, dev_got, rmat12, dev_rmat12):
    """FORA's guarantee (|est - pi| <= eps * pi wherever pi > delta = 1/n, failure probability 1/n) against the
    CPU ground truth, single-query and batched entry points, cost-model and fixed round counts."""
    for host, dev, srcs in ((got, dev_got, [63, 1, 5, 17, 42, 90]), (rmat12, dev_rmat12, live_sources(rmat12, 4, 3))):
        og = to_oracle(orc, host)
        exact = {s: og.power_method(s, A, 100) for s in srcs}
        delta = 1.0 / host.n
        for n_rounds in (0, 1, 3):
            for s in srcs:
                est, st = dev.fora_single_source(s, EPS, A, seed=3, n_rounds=n_rounds)
                # floor(omega * rsum) = 0 walks leaves (1 - alpha) of the residues undelivered, in the reference too
                # (Fora_Whole_Graph.java:112-113,123: omega_i = ceil(r / rsum * 0) = 0)
                assert abs(est.sum() - 1.0) < 1e-9 or (st.walks == 0 and abs(est.sum() + st.rsum - 1.0) < 1e-12)
                assert fora_bound_ok(est, exact[s], EPS, delta), (s, n_rounds)
        for tun in (pkg.tuning_default(), pkg.tuning_batch()):
            dev.set_tuning(tun)
            out, _, _, _, _, _ = dev.fora_batch_single_source(srcs, EPS, A, seed=3, fetch=True)
            for i, s in enumerate(srcs):
                assert fora_bound_ok(out[i], exact[s], EPS, delta), s
        dev.set_tuning(pkg.tuning_default())
    # a smaller eps tightens the estimate as the formulas say
    og = to_oracle(orc, got)
    pm = og.power_method(63, A, 100)
    e1, _ = dev_got.fora_single_source(63, 0.5, A, seed=4)
    e2, _ = dev_got.fora_single_source(63, 0.05, A, seed=4)
    assert orc.max_err(e2, pm) < orc.max_err(e1, pm) and orc.max_err(e2, pm) < 2e-3


# ------------------------------------------------------------------ (c) FORA top-k vs the power method's top-k
def test_fora_topk_vs_cpu_power_method(pkg, orc, got, dev_got, rmat12, dev_rmat12):
    for host, dev, srcs, k in ((got, dev_got, [63, 1, 5, 17, 42], 10), (rmat12, dev_rmat12, live_sources(rmat12, 4, 9), 32)):
        og = to_oracle(orc, host)
        for s in srcs:
            pm = og.power_method(s, A, 100)
            cnt, oids, ovals = orc.topk(pm, k, cap=host.n)
            v = np.sort(pm[pm > 0])[::-1]
            for eps in (EPS, 0.05):
                n_sel, ids, vals, est, _ = dev.fora_topk(s, eps, A, k, seed=5, cap=host.n, fetch=True)
                assert np.all(np.diff(vals) <= 0) and np.array_equal(est[ids], vals)
                kk = min(cnt, k)
                kth = ovals[kk - 1]
                # Fora_Topk's stopping rule (:175) bounds the relative error of the entries it reports by eps' = eps/2
                top = ids[:kk]
                assert np.all(np.abs(vals[:kk] - pm[top]) <= 0.5 * eps * np.maximum(pm[top], kth)), (s, eps)
                # set identity wherever the exact values around the k-th place are further apart than that error
                if cnt == k and v.size > k and (v[k - 1] - v[k]) > 2 * 0.5 * eps * v[k - 1]:
                    assert set(ids[:k].tolist()) == set(oids[:k].tolist()), (s, eps)
                assert orc.precision(ids[:kk], oids[:kk]) >= 0.8 and orc.ndcg(ids[:kk], oids[:kk], pm) >= 0.97


# ------------------------------------------------------------------ (d) the resumable top-k push, round by round
def test_fwdpush_topk_rounds_direct(pkg, orc, got, dev_got, rmat12, dev_rmat12):
    """pprhip_fwdpush_topk_reset / _round (Forward_Push.forward_push_topk, :144-250) called directly with
    Fora_Topk's sequence of thresholds, against the twin round by round and against the invariant."""
    for host, dev, srcs, k in ((got, dev_got, [63, 1, 71], 10), (rmat12, dev_rmat12, live_sources(rmat12, 2, 13), 32)):
        og = to_oracle(orc, host)
        conf = pkg.conf_topk(host.n, host.m, k, A)
        for s in srcs:
            dev.topk_push_reset(s, A)
            tw = og.topk_push(s, A, orc.SYNC)
            ff = og.topk_push(s, A, orc.FIFO)
            delta = 1.0 / k
            while True:
                min_rmax, rmax, _ = pkg.fora_topk_params(conf, EPS, delta)
                rsum, st = dev.topk_push_round(min_rmax, rmax)
                rs_t, st_t = tw.round(min_rmax, rmax)
                ff.round(min_rmax, rmax)
                p, r = dev.reserve(), dev.residue()
                assert np.max(np.abs(p - tw.reserve)) <= 1e-12 and np.max(np.abs(r - tw.residue)) <= 1e-12
                assert abs(rsum - rs_t) <= 1e-12 and st.levels == st_t.levels
                # against the reference's own order: same invariant, so the reserves differ by at most the residues
                assert np.max(np.abs(p - ff.reserve)) <= r.sum() + ff.residue.sum() + 1e-15
                od = np.diff(host.out_rp)
                if od[s] > 0:
                    assert abs(p.sum() + r.sum() - 1.0) < 1e-12
                    live = od > 0
                    # Forward_Push.java:173,226-237: whatever meets the round's threshold is pushed, unless it never
                    # received mass this round and was not parked (then it is below min_rmax; the scaled rmax of
                    # Fora_Topk.java:133 can fall below min_rmax in the last rounds)
                    assert np.all(r[live] / od[live] < max(rmax, min_rmax))
                    rf = ff.residue
                    assert np.all(rf[live] / od[live] < max(rmax, min_rmax))
                if delta <= 1.0 / host.n:
                    break
                delta = max(1.0 / host.n, delta / 4.0)
        with pytest.raises(pkg.PprhipError):  # a round needs a reset first (call-sequence error, PPRHIP_ERR_STATE)
            dev.forward_push(srcs[0], A, 1e-3)
            dev.topk_push_round(1e-9, 1e-3)


# ------------------------------------------------------------------ scale: CPU power method instead of the GPU's own
def test_fora_at_scale_vs_cpu_power_method(pkg, orc):
    """R-MAT 18 (the largest size the CPU power method finishes in seconds): FORA, single and batched, against the
    CPU ground truth under the (eps, delta) bound — not against the engine's own power method."""
    host = pkg.HostCsr.rmat(18, 16, seed=1)
    og = to_oracle(orc, host)
    srcs = live_sources(host, 3, 21)
    with pkg.Graph(host) as g:
        for s in srcs:
            pm = og.power_method(s, A, 100)
            est, st = g.fora_single_source(s, EPS, A, seed=7)
            big = pm > 1.0 / host.n
            assert np.all(np.abs(est[big] - pm[big]) <= EPS * pm[big])
            gpm, _ = g.power_method(s, A, 100)
            assert np.max(np.abs(gpm - pm)) <= 1e-12
            p, r, _, _ = g.forward_push(s, A, 1e-14)
            assert np.max(np.abs(p - pm)) <= TOL_SPEC
            n_sel, ids, _, _, _ = g.topk_select(32, cap=64)
            cnt, oids, _ = orc.topk(pm, 32, cap=64)
            if topk_gap_ok(pm, 32, host.m * 1e-14 + 1e-7):
                assert n_sel == cnt == 32 and set(ids.tolist()) == set(oids.tolist())
        g.set_tuning(pkg.tuning_batch())
        out, ids, vals, nsel, _, _ = g.fora_batch_single_source(srcs, EPS, A, seed=7, k=32, fetch=True)
        for i, s in enumerate(srcs):
            pm = og.power_method(s, A, 100)
            big = pm > 1.0 / host.n
            assert np.all(np.abs(out[i][big] - pm[big]) <= EPS * pm[big])
