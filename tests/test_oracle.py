"""CPU tests of the oracle (oracle/ppr_oracle.c): the restatement is pinned against everything the
reference publishes for this path — the thesis' two-node closed form (Dissertation.pdf p.13-14),
closed forms that follow from the same recurrence, the parameter formulas evaluated in
SURVEY.md §8(d), the Game-of-Thrones fixture's structure — and against the push invariant, which
ties the Java-faithful FIFO schedule to the frontier-synchronous twin the GPU is compared with.
The reference ships no golden vectors of its own ("parity unpinned", oracle/ppr_oracle.h).
"""
import math

import numpy as np
import pytest

from conftest import edges_to_host, to_oracle

A = 0.15


# ------------------------------------------------------------------ generator
def test_philox_known_answers(orc):
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in orc.philox([0, 0, 0, 0], [0, 0])] == ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in orc.philox([0xffffffff] * 4, [0xffffffff] * 2)] == [
        '0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in orc.philox([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                       [0xa4093822, 0x299f31d0])] == ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


# ------------------------------------------------------------------ parameters (Algo_Conf / FORA formulas)
def test_parameter_table(orc, pkg):
    # SURVEY.md §8(d) "Derived parameters" (natural log)
    rows = [(107, 352, 9.346e-3, 7.554e-4, 5.742e3), (1 << 20, 16 << 20, 9.537e-7, 2.122e-8, 1.526e8),
            (1 << 22, 16 << 22, 2.384e-7, 5.070e-9, 6.687e8), (1 << 24, 16 << 24, 5.960e-8, 1.216e-9, 2.907e9)]
    for n, m, delta, rmax0, omega in rows:
        for mod, conf in ((orc, None), (pkg, None)):
            if mod is orc:
                c = orc.Conf()
                orc.lib().orc_conf_fora_whole_graph(n, m, A, c)
                r, w = orc.fora_whole_params(c, 0.5)
            else:
                c = pkg.conf_whole_graph(n, m, A)
                r, w = pkg.fora_whole_params(c, 0.5)
            assert c.delta == pytest.approx(delta, rel=5e-4) and c.pfail == c.delta and c.rsum == 1.0
            assert r == pytest.approx(rmax0, rel=5e-4)
            assert w == pytest.approx(omega, rel=5e-4)  # the table is rounded to four digits
    # top-k, RMAT-22, k = 32: p_f' = 4.824e-15, min_rmax = 1.483e-9, round 1 rmax 9.667e-6 / omega 3.877e4
    c = pkg.conf_topk(1 << 22, 16 << 22, 32, A)
    co = orc.Conf()
    orc.lib().orc_conf_fora_topk(1 << 22, 16 << 22, 32, A, co)
    assert c.pfail == co.pfail == pytest.approx(4.824e-15, rel=2e-4)
    for f, cc in ((pkg.fora_topk_params, c), (orc.fora_topk_params, co)):
        min_rmax, rmax, omega = f(cc, 0.5, 1.0 / 32)
        assert min_rmax == pytest.approx(1.483e-9, rel=2e-4)
        assert rmax == pytest.approx(9.667e-6, rel=2e-3)
        assert omega == pytest.approx(3.877e4, rel=2e-4)
    # the product's formulas and the oracle's are written separately and must agree to the last bit
    assert pkg.fora_topk_params(c, 0.5, 1.0 / 512) == orc.fora_topk_params(co, 0.5, 1.0 / 512)
    # integer division in ln(n div k) (Algo_Conf.java:76)
    c7 = pkg.conf_topk(107, 352, 10, A)
    assert c7.pfail == pytest.approx(1.0 / 107 / 107 / math.log(10), rel=1e-12)


# ------------------------------------------------------------------ fixtures' structure
def test_got_fixture(got):
    od, idg = np.diff(got.out_rp), np.diff(got.in_rp)
    assert (got.n, got.m) == (107, 352)
    assert int((od == 0).sum()) == 36 and int((idg == 0).sum()) == 7
    assert int(od.max()) == 24 and got.names[int(od.argmax())] == "Tyrion"
    # no self loops, no duplicate or reciprocal edges
    e = {(v, int(u)) for v in range(got.n) for u in got.out_ci[got.out_rp[v]:got.out_rp[v + 1]]}
    assert len(e) == 352 and all(a != b for a, b in e) and all((b, a) not in e for a, b in e)
    # newest relationship first: Aemon's CSV rows are Grenn, Samwell -> stored Samwell, Grenn
    aemon = got.names.index("Aemon")
    assert [got.names[i] for i in got.out_ci[got.out_rp[aemon]:got.out_rp[aemon + 1]]] == ["Samwell", "Grenn"]


# ------------------------------------------------------------------ closed forms
def test_two_node_closed_form(orc, toy_graphs):
    """Dissertation p.13-14: V = {s, t}, E = {(s, t)}; t is a dead end that returns to s."""
    og = to_oracle(orc, toy_graphs["two_node"])
    ps, pt = A / (1 - (1 - A) ** 2), A * (1 - A) / (1 - (1 - A) ** 2)
    pm = og.power_method(0, A, 100)
    assert pm[0] == pytest.approx(ps, abs=1e-7) and pm[1] == pytest.approx(pt, abs=1e-7)
    for sch in (orc.FIFO, orc.SYNC):
        p, r, rsum, st = og.forward_push(0, A, 1e-13, sch)
        assert p[0] == pytest.approx(ps, abs=1e-12) and p[1] == pytest.approx(pt, abs=1e-12)
        assert p.sum() + r.sum() == pytest.approx(1.0, abs=1e-14)
    # dead-end source short-circuit (Forward_Push.java:72-76)
    p, r, rsum, st = og.forward_push(1, A, 1e-3, orc.FIFO)
    assert list(p) == [0.0, 1.0] and rsum == 0.0


def test_cycle_closed_form(orc, toy_graphs):
    og = to_oracle(orc, toy_graphs["cycle5"])
    exact = np.array([A * (1 - A) ** j / (1 - (1 - A) ** 5) for j in range(5)])
    for sch in (orc.FIFO, orc.SYNC):
        p, r, rsum, st = og.forward_push(0, A, 1e-14, sch)
        assert np.max(np.abs(p - exact)) < 1e-12
    pb, rb, st = og.backward_push(0, A, 1e-14, orc.FIFO)  # pi(v, 0): v is 5 - v steps before 0
    exact_b = np.array([exact[(5 - v) % 5] for v in range(5)])
    assert np.max(np.abs(pb - exact_b)) < 1e-12


def test_star_with_dead_end_leaves(orc, toy_graphs):
    og = to_oracle(orc, toy_graphs["star_dead_leaves"])
    # centre keeps alpha / (1 - (1-alpha)^2), every leaf a fifth of the rest
    centre = A / (1 - (1 - A) ** 2)
    for sch in (orc.FIFO, orc.SYNC):
        p, r, rsum, st = og.forward_push(0, A, 1e-14, sch)
        assert p[0] == pytest.approx(centre, abs=1e-12)
        assert np.allclose(p[1:], (1 - centre) / 5, atol=1e-12)


# ------------------------------------------------------------------ schedules and invariants
def exact_ppr_matrix(host, s, alpha):
    """alpha (I - (1-alpha) P_s)^-1 where dead-end rows of P_s point to the query source s: row v is
    the distribution of a walk started at v under the reference's dynamics for source s."""
    n = host.n
    P = np.zeros((n, n))
    for v in range(n):
        nb = host.out_ci[host.out_rp[v]:host.out_rp[v + 1]]
        if len(nb) == 0:
            P[v, s] = 1.0
        else:
            np.add.at(P[v], nb, 1.0 / len(nb))
    return alpha * np.linalg.inv(np.eye(n) - (1 - alpha) * P)


@pytest.mark.parametrize("rmax", [7.554e-4, 1e-5, 1e-9])
def test_push_invariant_both_schedules(orc, got, rmax):
    """pi(s, .) = reserve + sum_v residue(v) pi_s(v, .) for both schedules, with pi_s from a dense
    linear solve (independent of every oracle routine); the power method is checked against the
    same solve."""
    og = to_oracle(orc, got)
    od = np.diff(got.out_rp)
    for s in (0, 17, 42, 99):
        if od[s] == 0:
            for sch in (orc.FIFO, orc.SYNC):
                p, r, rsum, st = og.forward_push(s, A, rmax, sch)
                assert p[s] == 1.0 and p.sum() == 1.0 and rsum == 0.0
            continue
        Pi = exact_ppr_matrix(got, s, A)
        assert np.max(np.abs(og.power_method(s, A, 100) - Pi[s])) < 1e-7  # (1-alpha)^100 left undelivered
        for sch in (orc.FIFO, orc.SYNC):
            p, r, rsum, st = og.forward_push(s, A, rmax, sch)
            assert np.max(np.abs(p + r @ Pi - Pi[s])) < 1e-13
            assert np.all((od == 0) & (r == 0) | (od > 0) & (r / np.maximum(od, 1) < rmax))
            assert p.sum() + r.sum() == pytest.approx(1.0, abs=1e-12)


def test_fifo_rsum_quirk(orc, toy_graphs):
    """Forward_Push.java:140 updates rsum inside the loop but the dead-end `continue` (:114) skips
    it, so rsum can stay stale-high when the last pops are dead ends; the twin returns the exact sum."""
    og = to_oracle(orc, toy_graphs["two_node"])
    p, r, rsum_fifo, st = og.forward_push(0, A, 1e-12, orc.FIFO)
    p2, r2, rsum_sync, st2 = og.forward_push(0, A, 1e-12, orc.SYNC)
    assert rsum_sync == pytest.approx(r2.sum(), abs=0) and rsum_fifo >= r.sum()
    assert rsum_fifo == pytest.approx(r.sum() / (1 - A), rel=1e-9)  # one dead-end pop behind


def test_sync_levels_equal_fifo_fixed_point_small_rmax(orc, rmat12):
    og = to_oracle(orc, rmat12)
    od = np.diff(rmat12.out_rp)
    s = int(np.argmax(od > 2))
    pm = og.power_method(s, A, 300)
    for sch in (orc.FIFO, orc.SYNC):
        p, r, rsum, st = og.forward_push(s, A, 1e-13, sch)
        assert np.max(np.abs(p - pm)) < 1e-6  # north_star's 1e-6 L-inf bar against the exact vector


def test_topk_push_rounds(orc, got):
    og = to_oracle(orc, got)
    conf = og.conf_topk(10, A)
    for sch in (orc.FIFO, orc.SYNC):
        tp = og.topk_push(42, A, sch)
        delta, rs_prev = conf.delta, 1.0
        for _ in range(3):
            min_rmax, rmax, omega = orc.fora_topk_params(conf, 0.5, delta)
            rsum, st = tp.round(min_rmax, rmax)
            assert rsum <= rs_prev + 1e-15
            assert tp.reserve.sum() + tp.residue.sum() == pytest.approx(1.0, abs=1e-12)
            assert rsum == pytest.approx(tp.residue.sum(), abs=1e-12)
            od = np.diff(got.out_rp)
            live = od > 0
            assert np.all(tp.residue[live] / od[live] < rmax)
            rs_prev, delta = rsum, max(conf.min_delta, delta / 4)


# ------------------------------------------------------------------ walks
def test_walk_semantics(orc, toy_graphs, got):
    og = to_oracle(orc, toy_graphs["isolated_mix"])
    for s in (3, 4, 5):  # no out-edges: the walk returns its start (Monte_Carlo.java:70-72)
        assert og.random_walk(s, A, 1, 0, 7, False) == (s, 0)
        assert og.random_walk(s, A, 1, 0, 7, True) == (s, 0)
    og = to_oracle(orc, got)
    steps0 = [og.random_walk(17, A, 5, 0, i, False)[1] for i in range(20000)]
    steps1 = [og.random_walk(17, A, 5, 0, i, True)[1] for i in range(20000)]
    assert np.mean(steps0) == pytest.approx((1 - A) / A, rel=0.05)       # 5.67 (Dissertation p.16)
    assert np.mean(steps1) == pytest.approx(1 + (1 - A) / A, rel=0.05)   # forced first hop
    assert min(steps1) >= 1 and min(steps0) == 0
    # pure function of (seed, stream, start, index)
    assert og.random_walk(17, A, 5, 0, 123, True) == og.random_walk(17, A, 5, 0, 123, True)
    assert len({og.random_walk(17, A, 5, s, 123, True) for s in range(16)}) > 1


# ------------------------------------------------------------------ FORA
@pytest.mark.parametrize("schedule", [0, 1])
def test_fora_whole_guarantee_got(orc, got, schedule):
    """(eps, delta, p_f) guarantee against the power method: relative error <= eps where pi > delta."""
    og = to_oracle(orc, got)
    for s in (0, 17, 42):
        exact = og.power_method(s, A, 100)
        est, st = og.fora_whole(s, 0.5, A, seed=11, n_rounds=1, schedule=schedule)
        assert est.sum() == pytest.approx(1.0, abs=1e-9)
        big = exact > 1.0 / got.n
        assert np.all(np.abs(est[big] - exact[big]) <= 0.5 * exact[big])
        assert orc.max_err(est, exact) < 0.05
        if schedule == orc.SYNC:  # exact rsum: the per-node ceilings add up to at least the budget
            assert st.walks >= math.floor(st.omega * st.rsum)


def test_fora_rounds_trade_push_for_walks(orc, rmat12):
    og = to_oracle(orc, rmat12)
    od = np.diff(rmat12.out_rp)
    s = int(np.argmax(od > 2))
    prev_walks, prev_pushes = None, None
    for rounds in (1, 2, 3):
        est, st = og.fora_whole(s, 0.5, A, seed=1, n_rounds=rounds, schedule=orc.SYNC)
        if prev_walks is not None:
            assert st.walks < prev_walks and st.pops + st.dense_nodes > prev_pushes
        prev_walks, prev_pushes = st.walks, st.pops + st.dense_nodes
        assert st.rmax_final == pytest.approx(orc.fora_whole_params(og.conf_whole(A), 0.5)[0] / 2 ** (rounds - 1))
    est0, st0 = og.fora_whole(s, 0.5, A, seed=1, n_rounds=0, schedule=orc.SYNC)
    assert 1 <= st0.rounds <= 24


def test_threshold_rules_of_the_twin(orc, rmat12):
    """The three rules that stand in for the reference's wall-clock loop (cut rounds, halvings taken at once,
    a-priori start below rmax0) choose thresholds only: every choice is a power-of-two fraction of rmax0, the
    estimate keeps FORA's guarantee against the power method, and with the rules switched off the twin pushes
    at rmax0, rmax0/2, ... to the end of every round."""
    og = to_oracle(orc, rmat12)
    rmax0 = orc.fora_whole_params(og.conf_whole(A), 0.5)[0]
    od = np.diff(rmat12.out_rp)
    s = int(np.argmax(od > 2))
    exact = og.power_method(s, A, 100)
    big = exact > 1.0 / rmat12.n
    results = {}
    for name, patch in (("default", {}), ("batch", {"c_dense_edge_ns": 0.002, "c_dense_node_ns": 0.003, "dense_frac": 0.02}),
                        ("plain", {"halving_ratio": 1.0, "prior_levels": -1})):
        t = orc.tuning_default()
        for k, v in patch.items():
            setattr(t, k, v)
        est, st = og.fora_whole(s, 0.5, A, seed=1, n_rounds=0, schedule=orc.SYNC, tuning=t)
        results[name] = st
        frac = rmax0 / st.rmax_final
        assert abs(frac - 2 ** round(math.log2(frac))) < 1e-9 * frac          # a power-of-two fraction of rmax0
        assert abs(est.sum() - 1.0) < 1e-9 and np.all(np.abs(est[big] - exact[big]) <= 0.5 * exact[big])
        assert st.walks >= math.floor(st.omega * st.rsum)
    # a cheaper dense level (batch profile) can only push further: lower final threshold, fewer walks
    assert results["batch"].rmax_final <= results["default"].rmax_final
    assert results["batch"].walks <= results["default"].walks
    # rules off: one halving per round, so the final threshold is rmax0 / 2^(rounds - 1)
    assert results["plain"].rmax_final == pytest.approx(rmax0 / 2 ** (results["plain"].rounds - 1))
    # explicit round counts ignore the halving rules altogether
    est, st = og.fora_whole(s, 0.5, A, seed=1, n_rounds=3, schedule=orc.SYNC)
    assert st.rounds == 3 and st.rmax_final == pytest.approx(rmax0 / 4)


def test_fora_topk_precision_got(orc, got):
    og = to_oracle(orc, got)
    for s in (17, 42):
        exact = og.power_method(s, A, 100)
        _, gids, _ = orc.topk(exact, 10)
        for sch in (orc.FIFO, orc.SYNC):
            est, st = og.fora_topk(s, 0.5, A, 10, seed=4, schedule=sch)
            _, ids, _ = orc.topk(est, 10)
            assert orc.precision(ids, gids) >= 0.8
            assert orc.ndcg(ids, gids, exact) >= 0.95
            assert 1 <= st.rounds <= 3


# ------------------------------------------------------------------ top-k selection rule
def test_kth_and_topk_rule(orc):
    v = np.array([0.0, 0.3, 0.1, 0.3, 0.0, 0.2, 0.1])
    assert orc.kth_largest(v, 1) == 0.3 and orc.kth_largest(v, 3) == 0.2 and orc.kth_largest(v, 5) == 0.1
    assert orc.kth_largest(v, 6) is None  # only five entries exist (zeros are absent keys)
    cnt, ids, vals = orc.topk(v, 4)       # ties at the k-th value are all kept (Fora_Topk.java:194-197)
    assert cnt == 5 and list(ids) == [1, 3, 5, 2, 6]
    cnt, ids, vals = orc.topk(v, 9)
    assert cnt == 5
    cnt, ids, vals = orc.topk(v, 2)
    assert cnt == 2 and list(ids) == [1, 3]


# ------------------------------------------------------------------ backward search / all pair
def test_backward_push_matches_forward_columns(orc, got):
    og = to_oracle(orc, got)
    od = np.diff(got.out_rp)
    # On nodes that cannot reach a dead end, pi^b(v, t) approximates pi(v, t); check a closed subgraph-free bound
    # through the invariant pi(v,t) = reserve(v) + sum_u pi(v,u) r(u) on the cycle instead (exact there).
    t = 17
    for sch in (orc.FIFO, orc.SYNC):
        p, r, st = og.backward_push(t, A, 1e-6, sch)
        assert p[t] >= A and np.all(r <= 1e-6 + 1e-18)
        assert np.all(p >= 0) and np.all(r >= 0)
    # zero in-degree target: reserve = {t: 1.0} (Backward_Search.java:46-49; 1.0, not alpha)
    t0 = int(np.argmax(np.diff(got.in_rp) == 0))
    p, r, st = og.backward_push(t0, A, 1e-4, orc.FIFO)
    assert p[t0] == 1.0 and p.sum() == 1.0


def test_all_pair_structure(orc, got):
    og = to_oracle(orc, got)
    thr = 1e-3
    off, tg, vl = og.all_pair_backward(A, thr, -1)
    assert off[-1] == len(tg) and np.all(vl >= thr)
    for v in range(got.n):  # k < 0: target order
        seg = tg[off[v]:off[v + 1]]
        assert np.all(np.diff(seg) > 0)
    off5, tg5, vl5 = og.all_pair_backward(A, thr, 5)
    for v in range(got.n):
        full = vl[off[v]:off[v + 1]]
        keep = vl5[off5[v]:off5[v + 1]]
        assert np.all(np.diff(keep) <= 0)
        if len(full) >= 5:
            kth = np.sort(full)[::-1][4]
            assert len(keep) == int((full >= kth).sum())
        else:
            assert len(keep) == len(full)
    # FIFO and level-synchronous backward pushes agree to within the threshold's slack
    offf, tgf, vlf = og.all_pair_backward(A, thr, -1, schedule=orc.FIFO)
    assert abs(int(offf[-1]) - int(off[-1])) <= 0.1 * off[-1]


# ------------------------------------------------------------------ metrics (Gen_Util.computeError)
def test_metrics(orc):
    exact = np.array([0.5, 0.2, 0.2, 0.1, 0.0])
    est = np.array([0.45, 0.25, 0.1, 0.1, 0.1])
    assert orc.max_err(est, exact) == pytest.approx(0.1)  # node 4 is not in the ground-truth map
    assert orc.precision([0, 1, 4], [0, 1, 2]) == pytest.approx(2 / 3)
    z = sum((2 ** exact[g] - 1) / math.log(i + 2) / math.log(2) for i, g in enumerate([0, 1, 2]))
    d = sum((2 ** (exact[a] if a in (0, 1, 2) else 0.0) - 1) / math.log(i + 2) / math.log(2)
            for i, a in enumerate([0, 1, 4]))
    assert orc.ndcg([0, 1, 4], [0, 1, 2], exact) == pytest.approx(d / z)


# ------------------------------------------------------------------ committed Java-faithful vectors
def test_fifo_oracle_reproduces_committed_vectors(orc, got):
    """tests/golden/got_fifo_golden.json (generator: tests/golden/make_fifo_golden.py) freezes the FIFO schedule and
    the power method on the reference's dataset: the oracle must still produce them bit for bit, and the
    frontier-synchronous twin must stay within the invariant's bound of them."""
    import json
    import os
    from conftest import ROOT
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "got_fifo_golden.json")))
    un = lambda v: np.array([float.fromhex(x) for x in v])
    og = to_oracle(orc, got)
    assert (g["graph"]["n"], g["graph"]["m"]) == (got.n, got.m) and len(g["sources"]) == 3
    conf = og.conf_whole(A)
    rmax0, omega = orc.fora_whole_params(conf, 0.5)
    assert rmax0.hex() == g["fora_params"]["rmax0"] and omega.hex() == g["fora_params"]["omega"]
    for name, e in g["sources"].items():
        s = e["id"]
        assert got.names[s] == e["name"]
        assert np.array_equal(og.power_method(s, A, 100), un(e["power_method_100"]))
        for tag in ("rmax0", "1e-10"):
            rec = e["forward_push_" + tag]
            rmax = float.fromhex(rec["rmax"])
            p, r, rsum, st = og.forward_push(s, A, rmax, orc.FIFO)
            assert np.array_equal(p, un(rec["reserve"])) and np.array_equal(r, un(rec["residue"]))
            assert rsum.hex() == rec["rsum_field"] and st.pops == rec["pops"] and st.edge_pushes == rec["edge_pushes"]
            ps, rs, _, _ = og.forward_push(s, A, rmax, orc.SYNC)
            assert np.max(np.abs(ps - p)) <= rs.sum() + r.sum() + 1e-15
        p, r, st = og.backward_push(s, A, 1e-8, orc.FIFO)
        assert np.array_equal(p, un(e["backward_push_1e-8"]["reserve"])) and st.pops == e["backward_push_1e-8"]["pops"]
        est, st = og.fora_whole(s, 0.5, A, seed=3, n_rounds=1, schedule=orc.FIFO)
        rec = e["fora_whole_1round_seed3"]
        assert np.array_equal(est, un(rec["estimate"])) and st.walks == rec["walks"] and st.walk_steps == rec["walk_steps"]
        cnt, ids, _ = orc.topk(og.power_method(s, A, 100), 10, cap=got.n)
        assert cnt == e["power_method_top10"]["count"] and ids.tolist() == e["power_method_top10"]["ids"]


def test_topk_rounds_push_whatever_meets_the_threshold(orc, pkg, got, rmat12):
    """Forward_Push.forward_push_topk (:173,226-237) enqueues on the *new* residue and queue membership only, so after a
    round nothing at or above max(rmax, min_rmax) is left, in either schedule - also in the rounds whose scaled
    rmax (Fora_Topk.java:133) is below min_rmax, where a node can meet the threshold without having been parked
    (the frontier-synchronous twin once tested "crosses the threshold" and left such nodes behind)."""
    hit = 0
    for host, k, srcs in ((got, 10, [63, 1, 5, 17, 42]), (rmat12, 32, [350, 734, 969])):
        og = to_oracle(orc, host)
        od = np.diff(host.out_rp)
        live = od > 0
        conf = pkg.conf_topk(host.n, host.m, k, A)
        for s in srcs:
            for sch in (orc.SYNC, orc.FIFO):
                tp = og.topk_push(s, A, sch)
                delta = 1.0 / k
                while True:
                    min_rmax, rmax, _ = pkg.fora_topk_params(conf, 0.5, delta)
                    hit += rmax < min_rmax
                    tp.round(min_rmax, rmax)
                    r = tp.residue
                    assert np.all(r[live] / od[live] < max(rmax, min_rmax)), (s, sch, delta)
                    if od[s] > 0:
                        assert abs(tp.reserve.sum() + r.sum() - 1.0) < 1e-12
                    if delta <= 1.0 / host.n:
                        break
                    delta = max(1.0 / host.n, delta / 4.0)
    assert hit > 0  # the case is exercised (GOT's last round)


def test_index_column_check_accepts_the_oracle_index_and_rejects_a_changed_one(orc, got, rmat12):
    """bench.py's index_column_check (the All-Pair samples' self-check; the GPU tests run it against the oracle at full
    size) on CPU: the oracle's own index passes against the oracle's searches - twin exactly, FIFO under the bound both
    orders share, entries cut by the k rule accounted for - and an index with an entry removed, a value changed, an
    entry invented or a pair doubled does not (Base_Whole_Graph.java:76-92,112-163)."""
    from bench import index_column_check
    for host, thr, k, targets in ((got, 1e-3, 4, list(range(0, 107, 3))), (rmat12, 2e-3, 3, list(range(0, 4096, 37)))):
        og = to_oracle(orc, host)
        off, tg, vl = og.all_pair_backward(0.15, thr, k, schedule=orc.SYNC)
        col = lambda t: og.backward_push(t, 0.15, thr, orc.SYNC)[0]  # noqa: E731
        st = index_column_check(off, tg, vl, targets, col, thr, k)
        assert st["entries_checked"] > 0 and st["max_abs_diff"] == 0.0
        if host is rmat12:
            assert st["entries_cut_by_k_rule"] > 0        # k = 3 really cuts rows there
        st_f = index_column_check(off, tg, vl, targets, lambda t: og.backward_push(t, 0.15, thr, orc.FIFO)[0], thr, k,
                                  slack=thr)
        assert st_f["max_abs_diff"] <= thr
        # a changed index: pick an entry whose target is in the sample and whose row keeps fewer than k entries
        rows = np.repeat(np.arange(host.n), np.diff(off).astype(np.int64))
        cand = [i for i in range(len(tg)) if tg[i] in set(targets) and off[rows[i] + 1] - off[rows[i]] < k]
        assert cand
        i = cand[len(cand) // 2]
        v = int(rows[i])
        # (a) the entry removed
        off2 = off.copy()
        off2[v + 1:] -= 1
        with pytest.raises(AssertionError):
            index_column_check(off2, np.delete(tg, i), np.delete(vl, i), targets, col, thr, k)
        # (b) its value changed beyond the tolerance
        vl2 = vl.copy()
        vl2[i] *= 1.0 + 1e-6
        with pytest.raises(AssertionError):
            index_column_check(off, tg, vl2, targets, col, thr, k)
        # (c) an entry the search does not yield (a source whose reserve is zero there), (d) the pair doubled
        t = int(tg[i])
        zero_src = int(np.nonzero(col(t) == 0.0)[0][0])
        for src, val in ((zero_src, 2 * thr), (v, float(vl[i]))):
            at = int(off[src + 1])
            off3 = off.copy()
            off3[src + 1:] += 1
            with pytest.raises(AssertionError):
                index_column_check(off3, np.insert(tg, at, t), np.insert(vl, at, val), targets, col, thr, k)
