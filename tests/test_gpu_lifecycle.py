"""Lifecycle of handles on the GPU: whatever a handle, a result store, an index, a communicator or a query stream took
is back when it has been closed - device memory by hipMemGetInfo, host memory by the process's resident set.  A service
that lifts graphs and answers queries for days (the query stream's use) must not grow."""
import gc
import os

import numpy as np
import pytest

ALPHA = 0.15
pytestmark = pytest.mark.gpu


def _rss_mb():
    with open("/proc/self/statm") as f:
        return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / 1e6


def _one_life(pkg, host, srcs):
    """Every entry point once on a fresh handle, everything closed afterwards."""
    with pkg.Graph(host, device=0) as g:
        g.fora_single_source(int(srcs[0]), 0.5, ALPHA, seed=3)
        g.fora_topk(int(srcs[1]), 0.5, ALPHA, 8, seed=3, cap=16)
        g.forward_push(int(srcs[2]), ALPHA, 1e-6)
        g.backward_push(int(srcs[3]), ALPHA, 1e-4)
        g.power_method(int(srcs[4]), ALPHA, iters=10)
        g.monte_carlo(int(srcs[5]), 0.5, ALPHA, seed=3)
        g.set_tuning(pkg.tuning_batch())
        store = pkg.Results(g, 24)
        g.fora_batch_single_source(srcs[:24], 0.5, ALPHA, seed=5, k=8, keep=store, fetch=True)
        with pkg.QueryStream(g, 0.5, ALPHA, k=8) as qs:
            t1 = qs.submit(srcs[:20], 6, keep=store)
            t2 = qs.submit(srcs[20:24], 7, keep=store, keep_first=20)
            qs.wait(t2)
            qs.wait(t1)
        store.sum(3)
        store.close()
        g.set_tuning(pkg.tuning_default())
        g.fora_batch_topk(srcs[:20], 8, 0.5, ALPHA, seed=9)
        ix, _ = g.all_pair_backward(ALPHA, 1e-3, 8, 0, 4096)
        ix.arrays()
        ix.close()
        g.release(pkg.Graph.RELEASE_ALL_PAIR | pkg.Graph.RELEASE_BATCH)
        g.fora_batch_single_source(srcs[:5], 0.5, ALPHA, seed=5, k=4)  # workspaces come back after a release
    gc.collect()


def test_handles_give_everything_back(pkg, rmat15, rmat12):
    rng = np.random.default_rng(3)
    live = np.nonzero(np.diff(rmat15.out_rp) > 0)[0]
    srcs = live[rng.integers(0, live.size, 32)].astype(np.int32)
    with pkg.Graph(rmat12, device=0) as probe:  # a small handle that stays: pprhip_device_memory needs one
        _one_life(pkg, rmat15, srcs)  # first life: code objects, the runtime's own pools, allocator arenas
        _one_life(pkg, rmat15, srcs)
        free0, _ = probe.device_memory()
        rss0 = _rss_mb()
        for _ in range(4):
            _one_life(pkg, rmat15, srcs)
        free1, _ = probe.device_memory()
        rss1 = _rss_mb()
    print("four more lives: device %+.1f MB, host RSS %+.1f MB" % ((free0 - free1) / 1e6, rss1 - rss0))
    assert free0 - free1 <= 64 << 20, "device memory not returned: %.1f MB after four more lives" % ((free0 - free1) / 1e6)
    assert rss1 - rss0 <= 200.0, "host memory grew by %.0f MB over four lives" % (rss1 - rss0)


def test_stream_left_open_is_closed_with_its_graph(pkg, rmat12):
    """pprhip_graph_destroy on a handle whose query stream is still open ends the driver thread first (the thread uses the
    handle) and leaves the stream object to its owner: later calls on it fail cleanly, its close frees it; a new handle
    works."""
    live = np.nonzero(np.diff(rmat12.out_rp) > 0)[0][:8].astype(np.int32)
    g = pkg.Graph(rmat12, device=0)
    g.set_tuning(pkg.tuning_batch())
    qs = pkg.QueryStream(g, 0.5, ALPHA, k=4)
    tk = qs.submit(live, 5)
    qs.wait(tk)
    qs.submit(live, 6)  # still in flight or queued when the graph goes
    g.close()           # ends the driver first; the stream object stays its owner's
    with pytest.raises(pkg.PprhipError, match="graph has been destroyed"):
        qs.submit(live, 7)
    qs.close()          # only frees the object now
    with pkg.Graph(rmat12, device=0) as g2:
        est, _ = g2.fora_single_source(int(live[0]), 0.5, ALPHA, seed=3)
        assert abs(est.sum() - 1.0) < 1e-9


def test_stream_closed_with_a_submission_nobody_waited_for(pkg, rmat12):
    """The close runs every queued submission to its end and the driver writes their top-k blocks: the output arrays of
    a submission that was never waited for must live until the close has returned (QueryStream.close
    dropped them first, and the driver then wrote into freed numpy memory).  The blocks are checked after the close."""
    live = np.nonzero(np.diff(rmat12.out_rp) > 0)[0][:24].astype(np.int32)
    with pkg.Graph(rmat12, device=0) as g:
        g.set_tuning(pkg.tuning_batch())
        _, want, _, _, _, _ = g.fora_batch_single_source(live, 0.5, ALPHA, seed=5, k=4)
        qs = pkg.QueryStream(g, 0.5, ALPHA, k=4)
        tk = qs.submit(live, 5)
        held = qs._out[tk]  # what the library writes into; the test keeps a reference to look at it afterwards
        qs.close()          # no wait: the close finishes the block
        assert not qs._out
        assert np.array_equal(held[0], want), "the block of an un-waited submission was not finished by the close"
        with pkg.QueryStream(g, 0.5, ALPHA, k=4) as qs2:
            for s in range(3):
                qs2.submit(live, 5 + s)   # leaves the with-block with three blocks queued or in flight
        est, _ = g.fora_single_source(int(live[0]), 0.5, ALPHA, seed=3)
        assert abs(est.sum() - 1.0) < 1e-9


def test_kernel_timing_is_an_option(pkg, rmat15):
    """pprhip_set_kernel_timing: by default a call only counts its groups of launches (class_launches, class_bytes) and
    records no events between its kernels; with the option on the same call also reports class times; at level 2 only
    the dense sweeps' (what bench.py's timed region runs with).  Results are the same either way."""
    live = np.nonzero(np.diff(rmat15.out_rp) > 0)[0]
    src = int(live[5])
    with pkg.Graph(rmat15, device=0) as g:
        was = pkg.set_kernel_timing(False)
        try:
            est0, st0 = g.fora_single_source(src, 0.5, ALPHA, seed=3)
            assert sum(st0.class_launches) > 0 and st0.levels > 0
            assert st0.class_ms[1] == 0.0 and st0.class_ms[2] == 0.0 and st0.total_ms > 0.0
            assert pkg.set_kernel_timing(True) is False
            est1, st1 = g.fora_single_source(src, 0.5, ALPHA, seed=3)
            assert st1.class_ms[1] + st1.class_ms[2] > 0.0
            assert list(st1.class_launches) == list(st0.class_launches)
            assert np.max(np.abs(est0 - est1)) <= 1e-12
            assert pkg.set_kernel_timing(2) is True
            est2, st2 = g.fora_single_source(src, 0.5, ALPHA, seed=3)
            assert st2.dense_levels > 0 and st2.class_ms[1] > 0.0 and st2.class_ms[2] == 0.0   # sweeps timed, sparse levels counted
            assert list(st2.class_launches) == list(st0.class_launches) and np.max(np.abs(est0 - est2)) <= 1e-12
            assert pkg.set_kernel_timing(False) == 2
        finally:
            pkg.set_kernel_timing(was)


def test_workspace_pool_without_memory_falls_back(pkg, rmat15, monkeypatch):
    """The batch driver's extra workspaces (32 for the 16 columns) are an option of the device's memory: when one of them
    cannot be completed (injected: the 5th device allocation of the call fails, PPRHIP_FAIL_ALLOC_AFTER) the half-built
    workspace is dropped, the call runs with the workspaces there are and gives the same results; the next call builds
    the pool, and releasing the batch state returns all of it."""
    live = np.nonzero(np.diff(rmat15.out_rp) > 0)[0][:40].astype(np.int32)
    with pkg.Graph(rmat15, device=0) as g, pkg.Graph(rmat15, device=0) as ref:
        t = pkg.tuning_batch()
        g.set_tuning(t)
        ref.set_tuning(t)
        out0, _, _, _, pq0, _ = ref.fora_batch_single_source(live, 0.5, ALPHA, seed=4, fetch=True, per_query=True)
        g.fora_batch_single_source(live[:8], 0.5, ALPHA, seed=4, fetch=True)   # the 16 workspaces of a small call
        free_small, _ = g.device_memory()
        monkeypatch.setenv("PPRHIP_FAIL_ALLOC_AFTER", "5")
        out1, _, _, _, pq1, _ = g.fora_batch_single_source(live, 0.5, ALPHA, seed=4, fetch=True, per_query=True)
        monkeypatch.delenv("PPRHIP_FAIL_ALLOC_AFTER")
        free_after_fail, _ = g.device_memory()
        assert np.max(np.abs(out1 - out0)) <= 1e-12
        assert all(pq1[i].levels == pq0[i].levels and pq1[i].walks == pq0[i].walks for i in range(len(live)))
        assert free_small - free_after_fail < (64 << 20)           # nothing of the dropped workspace stays behind
        out2, _, _, _, _, _ = g.fora_batch_single_source(live, 0.5, ALPHA, seed=4, fetch=True)
        free_pool, _ = g.device_memory()
        assert np.max(np.abs(out2 - out0)) <= 1e-12
        assert free_after_fail - free_pool > 16 * 8 * 8 * rmat15.n  # the pool is there now (>= 8 vectors of n doubles each)
        g.release(g.RELEASE_BATCH)
        free_end, _ = g.device_memory()
        assert free_end >= free_small


def test_stream_failure_reaches_every_submission(pkg, rmat12, monkeypatch):
    """A failure inside the stream's driver thread (injected: PPRHIP_STREAM_FAULT_AT) ends every open submission and every
    later call with the driver's error instead of leaving a waiter blocked; the close reports it, frees the batch state,
    and the handle answers queries again afterwards - through a new stream too."""
    live = np.nonzero(np.diff(rmat12.out_rp) > 0)[0][:40].astype(np.int32)
    with pkg.Graph(rmat12, device=0) as g:
        g.set_tuning(pkg.tuning_batch())
        monkeypatch.setenv("PPRHIP_STREAM_FAULT_AT", "21")
        qs = pkg.QueryStream(g, 0.5, ALPHA, k=4)
        ids1, _, _, _ = qs.wait(qs.submit(live[:20], 5))  # queries 0 .. 19 of the stream
        assert (ids1[:, 0] >= 0).all()
        t2 = qs.submit(live[20:], 6)   # query 21 of the stream is never started: the stream fails with t2 in flight
        with pytest.raises(pkg.PprhipError, match="injected failure"):
            qs.wait(t2)
        with pytest.raises(pkg.PprhipError, match="injected failure"):
            qs.submit(live[:4], 7)
        with pytest.raises(pkg.PprhipError, match="injected failure"):
            qs.close()
        monkeypatch.delenv("PPRHIP_STREAM_FAULT_AT")
        _, ids, _, _, _, _ = g.fora_batch_single_source(live[:20], 0.5, ALPHA, seed=5, k=4)
        assert np.array_equal(ids[:, 0], ids1[:, 0])
        with pkg.QueryStream(g, 0.5, ALPHA, k=4) as qs2:
            ids2, _, _, _ = qs2.wait(qs2.submit(live[:20], 5))
        assert np.array_equal(ids2[:, 0], ids1[:, 0])


def test_store_and_communicator_outlive_their_graph(pkg, rmat12):
    """A result store or a communicator destroyed after the graph it was made for: their destructors need the device,
    not the graph (an order a garbage-collected host - the JVM, Python - easily produces)."""
    g = pkg.Graph(rmat12, device=0)
    store = pkg.Results(g, 4)
    comm = pkg.Comm(g, pkg.comm_unique_id(), 0, 1)
    g.close()
    store.close()
    comm.close()
    with pkg.Graph(rmat12, device=0) as g2:
        est, _ = g2.fora_single_source(int(np.nonzero(np.diff(rmat12.out_rp) > 0)[0][0]), 0.5, ALPHA, seed=3)
        assert abs(est.sum() - 1.0) < 1e-9
