"""GPU tests of the multi-GPU entry points behind the C ABI.

A gpurun box has one GPU, so the cases with several ranks put their graph replicas on that one device: the calls
then run one host thread per replica, shard queries / targets exactly as on n GPUs, partition the index entries by
owner of the source on the device and exchange them through the in-process transport (RCCL refuses two ranks on one
device).  The RCCL transport itself is exercised with a group of one rank (communicator, self send/recv of the
grouped exchange).  Results must be identical to the single-GPU entry points."""
import numpy as np
import pytest

from conftest import shared_graph, to_oracle

pytestmark = pytest.mark.gpu
A = 0.15


@pytest.fixture
def replicas(pkg, rmat12, dev_cache):
    def make():
        gs = [pkg.Graph(rmat12) for _ in range(3)]
        for g in gs:
            g.set_tuning(pkg.tuning_batch())
        return gs
    return shared_graph(dev_cache, pkg, "replicas", make)


def test_shard_ranges(pkg):
    for n, w in ((107, 4), (4096, 3), (10, 10), (1 << 22, 8)):
        r = [pkg.shard_target_range(i, w, n) for i in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
        assert max(e - b for b, e in r) - min(e - b for b, e in r) <= 1


@pytest.mark.parametrize("n_gpu", [1, 2, 3])
def test_fora_batch_over_replicas_equals_single_gpu(pkg, rmat12, replicas, n_gpu):
    rng = np.random.default_rng(5)
    srcs = rng.integers(0, rmat12.n, size=23).astype(np.int32)
    ids, vals, nsel, sts = pkg.fora_batch_multi(replicas[:n_gpu], srcs, 8, 0.5, A, seed=3)
    _, ids1, vals1, nsel1, _, _ = replicas[0].fora_batch_single_source(srcs, 0.5, A, seed=3, k=8)
    assert np.array_equal(ids, ids1) and np.array_equal(nsel, nsel1)
    assert np.max(np.abs(vals - vals1)) <= 1e-9            # same walks, sums in another order
    assert sum(st.walks for st in sts) > 0 and len(sts) == n_gpu
    if n_gpu > 1:
        assert all(st.levels > 0 for st in sts)            # every replica took part


@pytest.mark.parametrize("cut", ["auto", "work"])
@pytest.mark.parametrize("n_gpu,k", [(1, -1), (2, -1), (3, 4), (2, 0)])
def test_all_pair_over_replicas_equals_single_gpu(pkg, orc, rmat12, replicas, n_gpu, k, cut, monkeypatch):
    """(cut: the target ranges as the call decides them, and forced by work - PPRHIP_SHARD_CUT=work: the pilot's searches
    on rank 0's handle before its own share, ranges of any size and alignment)"""
    if cut == "work":
        monkeypatch.setenv("PPRHIP_SHARD_CUT", "work")
    thr = 2e-3
    ix, sts = pkg.all_pair_backward_multi(replicas[:n_gpu], A, thr, k)
    off, tg, vl = ix.arrays()
    ix1, _ = replicas[0].all_pair_backward(A, thr, k)
    off1, tg1, vl1 = ix1.arrays()
    assert np.array_equal(off, off1) and np.array_equal(tg, tg1) and np.max(np.abs(vl - vl1)) <= 1e-12
    # every entry was found by exactly one rank, and every rank's received bytes are whole 16-byte records
    assert sum(st.mc_sources for st in sts) >= len(tg) and all(st.select_bytes % 16 == 0 for st in sts)
    # and against the CPU oracle
    ooff, otg, ovl = to_oracle(orc, rmat12).all_pair_backward(A, thr, k, schedule=orc.SYNC)
    assert np.array_equal(off, ooff) and np.array_equal(tg, otg) and np.max(np.abs(vl - ovl)) <= 1e-12


def test_all_pair_tier3_entries_join_the_exchange(pkg, rmat12, replicas, monkeypatch):
    monkeypatch.setenv("PPRHIP_APBS_TIER", "3")  # whole-vector searches: their entries reach the store from the host
    ix, _ = pkg.all_pair_backward_multi(replicas[:2], A, 5e-3, 4)
    monkeypatch.delenv("PPRHIP_APBS_TIER")
    ix1, _ = replicas[0].all_pair_backward(A, 5e-3, 4)
    a, b = ix.arrays(), ix1.arrays()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.max(np.abs(a[2] - b[2])) <= 1e-12


def test_rccl_group_of_one(pkg, rmat12, replicas):
    """The RCCL transport: unique id, communicator, grouped send/recv to self, through the sharded entry points."""
    uid = pkg.comm_unique_id()
    assert len(uid) == pkg.COMM_ID_BYTES
    c = pkg.Comm(replicas[0], uid, 0, 1)
    try:
        own, st = c.all_pair_backward_sharded(A, 2e-3, 4)
        ix1, _ = replicas[0].all_pair_backward(A, 2e-3, 4)
        a, b = own.arrays(), ix1.arrays()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.max(np.abs(a[2] - b[2])) <= 1e-12
        assert st.select_bytes == 16 * st.mc_sources       # everything it found came back through the exchange
        ids = np.arange(12, dtype=np.int32).reshape(3, 4)
        vals = np.linspace(1.0, 0.1, 12).reshape(3, 4)
        ri, rv = c.topk_gather(ids, vals, rows_max=5)
        assert ri.shape == (1, 5, 4) and np.array_equal(ri[0, :3], ids) and np.all(ri[0, 3:] == -1)
        assert np.array_equal(rv[0, :3], vals) and np.all(rv[0, 3:] == 0.0)
    finally:
        c.close()


@pytest.mark.parametrize("where", ["search", "partition", "exchange"])
@pytest.mark.parametrize("bad", [0, 2])
def test_all_pair_rank_failure_reaches_every_rank(pkg, rmat12, replicas, where, bad, monkeypatch):
    """A rank that fails before or inside the exchange must not leave its peers waiting: it takes part in the
    exchange with an error mark, every rank returns, the call reports the failing rank - and the handles are usable
    again afterwards (PPRHIP_FAULT_* are the library's test switches)."""
    monkeypatch.setenv("PPRHIP_FAULT_RANK", str(bad))
    monkeypatch.setenv("PPRHIP_FAULT_AT", where)
    with pytest.raises(pkg.PprhipError) as ei:
        pkg.all_pair_backward_multi(replicas[:3], A, 2e-3, 4)
    assert "injected fault on rank %d" % bad in str(ei.value)
    monkeypatch.delenv("PPRHIP_FAULT_RANK")
    monkeypatch.delenv("PPRHIP_FAULT_AT")
    ix, _ = pkg.all_pair_backward_multi(replicas[:3], A, 2e-3, 4)
    ix1, _ = replicas[0].all_pair_backward(A, 2e-3, 4)
    a, b = ix.arrays(), ix1.arrays()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.max(np.abs(a[2] - b[2])) <= 1e-12


@pytest.mark.parametrize("where", ["search", "gather"])
def test_fora_batch_rank_failure_reaches_every_rank(pkg, rmat12, replicas, where, monkeypatch):
    srcs = np.arange(10, 21, dtype=np.int32)
    monkeypatch.setenv("PPRHIP_FAULT_RANK", "1")
    monkeypatch.setenv("PPRHIP_FAULT_AT", where)
    with pytest.raises(pkg.PprhipError) as ei:
        pkg.fora_batch_multi(replicas[:3], srcs, 4, 0.5, A, seed=3)
    assert "injected fault on rank 1" in str(ei.value)
    monkeypatch.delenv("PPRHIP_FAULT_RANK")
    monkeypatch.delenv("PPRHIP_FAULT_AT")
    ids, vals, nsel, _ = pkg.fora_batch_multi(replicas[:3], srcs, 4, 0.5, A, seed=3)
    _, ids1, _, nsel1, _, _ = replicas[0].fora_batch_single_source(srcs, 0.5, A, seed=3, k=4)
    assert np.array_equal(ids, ids1) and np.array_equal(nsel, nsel1)


def test_rccl_rank_failure_group_of_one(pkg, rmat12, replicas, monkeypatch):
    """The same protocol on the RCCL transport (a group of one: the rank posts the error mark to itself)."""
    c = pkg.Comm(replicas[0], pkg.comm_unique_id(), 0, 1)
    try:
        monkeypatch.setenv("PPRHIP_FAULT_RANK", "0")
        monkeypatch.setenv("PPRHIP_FAULT_AT", "partition")
        with pytest.raises(pkg.PprhipError) as ei:
            c.all_pair_backward_sharded(A, 2e-3, 4)
        assert "injected fault on rank 0" in str(ei.value)
        monkeypatch.delenv("PPRHIP_FAULT_RANK")
        own, _ = c.all_pair_backward_sharded(A, 2e-3, 4)       # the communicator survived a failure announced in time
        ix1, _ = replicas[0].all_pair_backward(A, 2e-3, 4)
        assert np.array_equal(own.arrays()[1], ix1.arrays()[1])
        c.abort()                                               # leaving the group: later collectives fail at once
        with pytest.raises(pkg.PprhipError):
            c.all_pair_backward_sharded(A, 2e-3, 4)
    finally:
        c.close()


def test_rccl_branch_with_several_ranks_on_a_test_double(tmp_path):
    """comm.cpp's RCCL branch (size exchange with the error sentinel, payload groups, the top-k gather, failing and
    leaving ranks) with 2 and 3 ranks on the one GPU of this box: in a child process, the library binds
    tests/fixtures/fake_rccl.cpp - a test double, NOT RCCL (ranks are threads, a send / receive pair is a device
    copy) - through PPRHIP_RCCL_LIB.  What it cannot show is the fabric; what it does run is every line of ours
    around the dozen RCCL calls, which no multi-GPU box has executed yet."""
    import os
    import shutil
    import subprocess
    import sys
    from conftest import ROOT
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    lib = str(tmp_path / "libfake_rccl.so")
    subprocess.run([hipcc, "-O1", "-std=c++17", "-shared", "-fPIC", "-o", lib,
                    os.path.join(ROOT, "tests", "fixtures", "fake_rccl.cpp")], check=True, timeout=300)
    env = dict(os.environ, PPRHIP_RCCL_LIB=lib, PPRHIP_COMM_TIMEOUT_S="60", FAKE_RCCL_TIMEOUT_S="20")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fixtures", "rccl_double_run.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-2000:], r.stderr[-3000:])


def test_multi_argument_errors(pkg, rmat12, replicas, got):
    with pytest.raises(pkg.PprhipError):          # the same handle twice
        pkg.fora_batch_multi([replicas[0], replicas[0]], [1, 2], 4, 0.5, A, seed=1)
    other = pkg.Graph(got)
    try:
        with pytest.raises(pkg.PprhipError):      # replicas of different graphs
            pkg.all_pair_backward_multi([replicas[0], other], A, 1e-3, 4)
    finally:
        other.close()


@pytest.mark.timeout(900)
def test_work_weighted_target_cut_balances_a_degree_sorted_store(pkg, monkeypatch):
    """Eight ranks, a graph whose ids follow its in-degrees (what the import of a degree-sorted dump leaves in a store:
    Base_Whole_Graph.java:76-92 walks the targets in id order).  Equal counts of contiguous targets give rank 0 every hub;
    the work-weighted cut (pprhip_shard_target_cuts: a pilot measures the hubs' searches, the others are estimated from
    their in-degree) must leave every rank within 15 % of the mean of the edge pushes - and the merged index must be the
    single-GPU one either way.  The eight replicas share the box's one GPU (in-process transport)."""
    base = pkg.HostCsr.rmat(18, 16, seed=1)
    n = base.n
    ind = np.diff(base.in_rp).astype(np.int64)
    rank_of = np.empty(n, dtype=np.int64)
    rank_of[np.argsort(-ind, kind="stable")] = np.arange(n)            # new id = in-degree rank
    src = rank_of[np.repeat(np.arange(n), np.diff(base.out_rp))].astype(np.int32)
    dst = rank_of[base.out_ci].astype(np.int32)
    host = pkg.HostCsr(n, src, dst, False)
    assert np.all(np.diff(np.diff(host.in_rp).astype(np.int64)) <= 0)   # ids are in in-degree order
    thr, k, W = 1e-3, 16, 8
    gs = [pkg.Graph(host) for _ in range(W)]
    try:
        cuts_eq, skew = pkg.shard_target_cuts(gs[0], W, A, thr, pkg.CUT_EQUAL)
        cuts_w, _ = pkg.shard_target_cuts(gs[0], W, A, thr, pkg.CUT_BY_WORK)
        cuts_auto, _ = pkg.shard_target_cuts(gs[0], W, A, thr, pkg.CUT_AUTO)
        assert skew > 2.0 and np.array_equal(cuts_auto, cuts_w) and not np.array_equal(cuts_w, cuts_eq)
        assert cuts_w[0] == 0 and cuts_w[-1] == n and np.all(np.diff(cuts_w.astype(np.int64)) > 0)
        ix1, st1 = gs[0].all_pair_backward(A, thr, k)
        ref = [x.copy() for x in ix1.arrays()]
        ix1.close()
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ref[0]).astype(np.int64))
        o_ref = np.argsort(rows * n + ref[1], kind="stable")

        def same_index(a):
            # the same rows with the same entries; inside a row, entries whose values differ in the last bits (the
            # atomics' order) may swap places in the value order: compare by (row, target)
            if not (np.array_equal(a[0], ref[0]) and len(a[1]) == len(ref[1])):
                return False
            o = np.argsort(rows * n + a[1], kind="stable")
            return np.array_equal(a[1][o], ref[1][o_ref]) and np.max(np.abs(a[2][o] - ref[2][o_ref])) <= 1e-12

        # the ranges of the weighted cut one after the other on ONE handle, merged: a range of a few hub targets yields
        # more entries than the record buffer of a small range used to hold (round 5: the hubs' searches were repeated
        # a thousand times and then dropped without an error)
        shards = [gs[0].all_pair_backward(A, thr, -1, int(cuts_w[r]), int(cuts_w[r + 1]))[0] for r in range(W)]
        assert len(shards[0].arrays()[1]) > (1 << 16)
        merged = pkg.merge_indexes(shards, k)
        assert same_index(merged.arrays()), "ranges of the weighted cut, merged, are not the full index"
        for x in shards + [merged]:
            x.close()
        shares = {}
        for mode in ("count", "work"):
            monkeypatch.setenv("PPRHIP_SHARD_CUT", mode)
            ix, sts = pkg.all_pair_backward_multi(gs, A, thr, k)
            assert same_index(ix.arrays()), "the index of eight ranks (cut by %s) is not the single-GPU one" % mode
            ix.close()
            # a rank's work: edges pushed by the LDS and dense tiers + edges swept by its whole-vector searches
            e = np.array([st.edge_pushes + st.dense_edges for st in sts], dtype=np.float64)
            print(mode, "edge pushes", [int(st.edge_pushes) for st in sts], "swept", [int(st.dense_edges) for st in sts],
                  "whole-vector searches", [int(st.xl_targets) for st in sts], "single GPU:", st1.edge_pushes, st1.dense_edges)
            shares[mode] = e / e.mean()
        monkeypatch.delenv("PPRHIP_SHARD_CUT")
        print("edge pushes per rank / mean: equal counts %s, by work %s" % (np.round(shares["count"], 2), np.round(shares["work"], 2)))
        assert shares["count"].max() > 3.0                                       # rank 0 holds the hubs
        assert shares["work"].max() <= 1.15 and shares["work"].min() >= 0.85
        # the default decides by the modelled skew: here by work
        ix, sts = pkg.all_pair_backward_multi(gs, A, thr, k)
        e = np.array([st.edge_pushes + st.dense_edges for st in sts], dtype=np.float64)
        assert (e / e.mean()).max() <= 1.15
        ix.close()
    finally:
        for g in gs:
            g.close()


def test_work_cut_on_a_tiny_graph_with_more_ranks_than_hubs(pkg, orc, got, monkeypatch):
    """Game of Thrones (107 nodes) on eight ranks, cut by work: ranges of a handful of targets, possibly none - the
    merged index is the single-GPU one, and the cuts tile [0, n)."""
    gs = [pkg.Graph(got) for _ in range(8)]
    try:
        monkeypatch.setenv("PPRHIP_SHARD_CUT", "work")
        cuts, _ = pkg.shard_target_cuts(gs[0], 8, A, 1e-2, pkg.CUT_BY_WORK)
        assert cuts[0] == 0 and cuts[-1] == got.n and np.all(np.diff(cuts.astype(np.int64)) >= 0)
        ix, sts = pkg.all_pair_backward_multi(gs, A, 1e-2, 5)
        ix1, _ = gs[0].all_pair_backward(A, 1e-2, 5)
        a, b = ix.arrays(), ix1.arrays()
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.max(np.abs(a[2] - b[2])) <= 1e-12
        ooff, otg, ovl = to_oracle(orc, got).all_pair_backward(A, 1e-2, 5, schedule=orc.SYNC)
        assert np.array_equal(a[0], ooff) and np.array_equal(a[1], otg) and np.max(np.abs(a[2] - ovl)) <= 1e-12
    finally:
        for g in gs:
            g.close()


@pytest.mark.timeout(600)
def test_work_weighted_cut_with_empty_target_ranges(pkg, monkeypatch):
    """one target whose search outweighs total / W - a star: every node points at node 0, and a short chain
    behind the last node so that a few other searches find something - makes the work-weighted cut place several cuts at
    the same id: some of the eight ranks get an EMPTY target range, collect nothing and still take part in the exchange.
    The merged index must be the single-GPU one, every rank returns."""
    n = 3000
    src = np.concatenate([np.arange(1, n), np.arange(n - 40, n - 1)]).astype(np.int32)
    dst = np.concatenate([np.zeros(n - 1, dtype=np.int64), np.arange(n - 39, n)]).astype(np.int32)
    host = pkg.HostCsr(n, src, dst, False)
    thr, k, W = 1e-4, 8, 8
    gs = [pkg.Graph(host) for _ in range(W)]
    try:
        cuts_w, _ = pkg.shard_target_cuts(gs[0], W, A, thr, pkg.CUT_BY_WORK)
        widths = np.diff(cuts_w.astype(np.int64))
        assert cuts_w[0] == 0 and cuts_w[-1] == n and np.all(widths >= 0)
        assert (widths == 0).any(), "the star's hub should leave at least one rank without targets: %s" % cuts_w
        ix1, _ = gs[0].all_pair_backward(A, thr, k)
        ref = [x.copy() for x in ix1.arrays()]
        ix1.close()
        monkeypatch.setenv("PPRHIP_SHARD_CUT", "work")
        ix, sts = pkg.all_pair_backward_multi(gs, A, thr, k)
        got = ix.arrays()
        assert np.array_equal(got[0], ref[0]) and len(got[1]) == len(ref[1])
        rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ref[0]).astype(np.int64))
        o, o_ref = np.argsort(rows * n + got[1], kind="stable"), np.argsort(rows * n + ref[1], kind="stable")
        assert np.array_equal(got[1][o], ref[1][o_ref]) and np.max(np.abs(got[2][o] - ref[2][o_ref])) <= 1e-12
        ix.close()
        assert len(sts) == W
    finally:
        for g in gs:
            g.close()
