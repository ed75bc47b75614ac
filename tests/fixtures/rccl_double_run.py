"""Child process of tests/test_gpu_multi.py::test_rccl_branch_with_several_ranks_on_a_test_double.

PPRHIP_RCCL_LIB points at the test double built from tests/fixtures/fake_rccl.cpp (NOT RCCL: ranks are threads whose
graph replicas share the one GPU, a send / receive pair is a device copy).  Every rank is a Python thread with a
communicator of its own (pprhip_comm_create), as one process per GPU would have; the calls below are the library's
RCCL branch: size exchange with the error sentinel, payload groups with self send / receive, the gather of the top-k
blocks, a rank that fails before the exchange, a rank that leaves the group.  Prints "ok" at the end."""
import importlib
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
A = 0.15
assert os.environ.get("PPRHIP_RCCL_LIB"), "the test double's path"


def on_ranks(world, fn):
    """fn(rank) on one thread per rank; returns the results, or raises the ranks' errors together."""
    out, err = [None] * world, [None] * world

    def run(r):
        try:
            out[r] = fn(r)
        except Exception as e:  # noqa: BLE001
            err[r] = e

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
        assert not t.is_alive(), "a rank never returned"
    return out, err


host = pkg.HostCsr.rmat(12, 16, seed=1)
graphs = [pkg.Graph(host) for _ in range(3)]
for g in graphs:
    g.set_tuning(pkg.tuning_batch())
ref, _ = graphs[0].all_pair_backward(A, 2e-3, 4)
roff, rtg, rvl = [x.copy() for x in ref.arrays()]
ref.close()

for world in (2, 3):
    uid = pkg.comm_unique_id()
    comms, err = on_ranks(world, lambda r: pkg.Comm(graphs[r], uid, r, world))
    assert not any(err), err
    # ---- All-Pair: every rank ends up with the finished rows of the sources it owns
    res, err = on_ranks(world, lambda r: comms[r].all_pair_backward_sharded(A, 2e-3, 4))
    assert not any(err), err
    got_entries = 0
    for r in range(world):
        own, st = res[r]
        off, tg, vl = own.arrays()
        lo, hi = pkg.shard_target_range(r, world, host.n)
        # rows outside the rank's range are empty, rows inside equal the single-GPU index
        assert off[lo] == 0 and off[hi] == off[-1]
        a, b = roff[lo], roff[hi]
        assert np.array_equal(off[lo:hi + 1] - off[lo], roff[lo:hi + 1] - a)
        assert np.array_equal(tg, rtg[a:b]) and np.max(np.abs(vl - rvl[a:b]), initial=0.0) <= 1e-12
        assert st.select_bytes % 16 == 0
        got_entries += len(tg)
        own.close()
    assert got_entries == len(rtg)
    # ---- the gather of the top-k blocks: ragged row counts, padded with -1 / 0
    k = 4
    blocks = [(np.arange(r * 100, r * 100 + (r + 2) * k, dtype=np.int32).reshape(r + 2, k),
               np.linspace(1.0, 0.5, (r + 2) * k).reshape(r + 2, k) + r) for r in range(world)]
    res, err = on_ranks(world, lambda r: comms[r].topk_gather(blocks[r][0], blocks[r][1], rows_max=world + 1))
    assert not any(err), err
    ri, rv = res[0]
    assert all(x is None for x in res[1:])
    for r in range(world):
        assert np.array_equal(ri[r, :r + 2], blocks[r][0]) and np.all(ri[r, r + 2:] == -1)
        assert np.array_equal(rv[r, :r + 2], blocks[r][1]) and np.all(rv[r, r + 2:] == 0.0)
    # ---- a rank that fails before the exchange tells every peer; the group is usable afterwards
    for where in ("search", "partition", "exchange"):
        os.environ["PPRHIP_FAULT_RANK"] = str(world - 1)
        os.environ["PPRHIP_FAULT_AT"] = where
        res, err = on_ranks(world, lambda r: comms[r].all_pair_backward_sharded(A, 2e-3, 4))
        del os.environ["PPRHIP_FAULT_RANK"], os.environ["PPRHIP_FAULT_AT"]
        assert all(isinstance(e, pkg.PprhipError) for e in err), (where, err)
        assert "injected fault on rank %d" % (world - 1) in str(err[world - 1])
        if where == "exchange":  # the failing rank aborted its communicator: a new group for what follows
            for c in comms:
                c.close()
            uid = pkg.comm_unique_id()
            comms, err = on_ranks(world, lambda r: pkg.Comm(graphs[r], uid, r, world))
            assert not any(err), err
        res, err = on_ranks(world, lambda r: comms[r].all_pair_backward_sharded(A, 2e-3, 4))
        assert not any(err), (where, err)
        assert sum(len(x[0].arrays()[1]) for x in res) == len(rtg)
        for x in res:
            x[0].close()
    # ---- a rank that leaves the group: its peers' collective ends with an error instead of waiting
    comms[0].abort()
    res, err = on_ranks(world, lambda r: comms[r].all_pair_backward_sharded(A, 2e-3, 4))
    assert all(isinstance(e, pkg.PprhipError) for e in err), err
    for c in comms:
        c.close()

# ---- the one-process entry points (one host thread per replica, communicators from ncclCommInitAll) on the same branch
os.environ["PPRHIP_FORCE_RCCL"] = "1"
for world in (2, 3):
    ix, sts = pkg.all_pair_backward_multi(graphs[:world], A, 2e-3, 4)
    off, tg, vl = ix.arrays()
    assert np.array_equal(off, roff) and np.array_equal(tg, rtg) and np.max(np.abs(vl - rvl)) <= 1e-12
    assert all(st.select_bytes % 16 == 0 for st in sts)
    ix.close()
    srcs = np.random.default_rng(5).integers(0, host.n, size=23).astype(np.int32)
    ids, vals, nsel, sts = pkg.fora_batch_multi(graphs[:world], srcs, 8, 0.5, A, seed=3)
    _, ids1, vals1, nsel1, _, _ = graphs[0].fora_batch_single_source(srcs, 0.5, A, seed=3, k=8)
    assert np.array_equal(ids, ids1) and np.array_equal(nsel, nsel1) and np.max(np.abs(vals - vals1)) <= 1e-9
    os.environ["PPRHIP_FAULT_RANK"] = "1"
    os.environ["PPRHIP_FAULT_AT"] = "gather"
    try:
        pkg.fora_batch_multi(graphs[:world], srcs, 8, 0.5, A, seed=3)
        raise AssertionError("the injected fault was not reported")
    except pkg.PprhipError as e:
        assert "injected fault on rank 1" in str(e)
    del os.environ["PPRHIP_FAULT_RANK"], os.environ["PPRHIP_FAULT_AT"]
del os.environ["PPRHIP_FORCE_RCCL"]

for g in graphs:
    g.close()
print("ok")
