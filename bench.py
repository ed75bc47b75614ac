#!/usr/bin/env python3
"""bench.py — single-source FORA queries/sec on a synthetic R-MAT graph (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N = 1): R-MAT scale 22 (n = 4 194 304, m = 67 108 864, generator seed 1), single-source
FORA with alpha = 0.15, eps = 0.5; a *step* is one batch of `--queries-per-step` sources drawn
uniformly from [0, n) with seed 2 (as Gen_Util.getQueryNodes does, so dead-end sources, which
short-circuit, are in the mix), handed to pprhip_fora_batch_single_source: the queries are
independent single-source computations, 16 of them in flight at a time, and every query's top-32 is
selected on the device.  The graph is lifted into HBM once before the timed region; the PPR
vectors stay in HBM.  `value` = queries / second over all ranks.  (`--mode single` runs the same
queries one after another through pprhip_fora_single_source.)

N > 1 (one process per GPU, launched by torch.distributed.run): the CSR is replicated, every rank
runs its own batch per step (weak scaling, no data-path collective) and the per-step top-k blocks
are gathered to rank 0 over RCCL.

Extra objects on the JSON line: `roofline` (dominant kernel, HIP-event time on the engine's
stream, algorithmic bytes from DESIGN.md's byte model) and `cpu_baseline` (the CPU oracle's
clock-driven FIFO FORA, one core, bounded sample; rank 0 at N = 1 only); after the timed region,
at N = 1, also `one_query_at_a_time` (the same queries through the single-query entry point)
`topk_sample` (FORA top-32, batched and one at a time) and `all_pair_sample` (All-Pair-Backward-Search on
2^18 targets of the same graph).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALPHA = 0.15
EPS = 0.5
TOPK = 32
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--queries-per-step", type=int, default=128)
    ap.add_argument("--mode", choices=["batch", "single"], default="batch")
    ap.add_argument("--rounds", type=int, default=0, help="FORA threshold rounds (0 = cost model)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the one-query-at-a-time and All-Pair samples")
    ap.add_argument("--cpu-walk-divisor", type=int, default=16)
    ap.add_argument("--tuning", default="", help="cost-model overrides, e.g. c_dense_edge_ns=0.002,max_rounds=30")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    import torch  # first: libpprhip.so then binds to the HIP runtime torch already loaded
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the engine has no CPU fallback)")
    # one process per GPU; the modulo only matters when a launch is rehearsed on fewer devices (gloo)
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    xdev = "cuda" if args.backend == "nccl" else "cpu"  # where the gathered top-k blocks live
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")

    # ---- graph lift (outside the timed region)
    t0 = time.time()
    host = pkg.HostCsr.rmat(args.scale, 16, seed=1)
    t_gen = time.time() - t0
    t0 = time.time()
    g = pkg.Graph(host, device=local_rank)
    t_lift = time.time() - t0
    conf = pkg.conf_whole_graph(host.n, host.m, ALPHA)
    tuning = pkg.tuning_batch() if args.mode == "batch" else pkg.tuning_default()
    for kv in filter(None, args.tuning.split(",")):
        key, val = kv.split("=")
        setattr(tuning, key, type(getattr(tuning, key))(float(val)))
    g.set_tuning(tuning)
    outdeg = np.diff(host.out_rp)

    q = args.queries_per_step
    total_steps = args.warmup + args.steps
    rng = np.random.default_rng(2 + 7919 * rank)
    srcs = rng.integers(0, host.n, size=(total_steps, q)).astype(np.int32)

    ids_blk = torch.empty((q, TOPK), dtype=torch.int32, device=xdev)
    vals_blk = torch.empty((q, TOPK), dtype=torch.float64, device=xdev)
    gather_ids = [torch.empty_like(ids_blk) for _ in range(world)] if (world > 1 and rank == 0) else None
    gather_vals = [torch.empty_like(vals_blk) for _ in range(world)] if (world > 1 and rank == 0) else None

    acc = {"class_ms": [0.0] * 8, "class_bytes": [0] * 8, "class_launches": [0] * 8, "rounds": 0, "live": 0,
           "walks": 0, "walk_steps": 0, "levels": 0, "dense_levels": 0, "dense_edges": 0, "push_ms": 0.0, "mc_ms": 0.0}

    def record_stats(st, live):
        for c in range(8):
            acc["class_ms"][c] += st.class_ms[c]
            acc["class_bytes"][c] += st.class_bytes[c]
            acc["class_launches"][c] += st.class_launches[c]
        acc["rounds"] += st.rounds
        acc["live"] += live
        acc["walks"] += st.walks
        acc["walk_steps"] += st.walk_steps
        acc["levels"] += st.levels
        acc["dense_levels"] += st.dense_levels
        acc["dense_edges"] += st.dense_edges
        acc["push_ms"] += st.push_ms
        acc["mc_ms"] += st.mc_ms

    def run_step(i, record):
        if args.mode == "batch":
            _, ids, vals, _, _, st = g.fora_batch_single_source(srcs[i], EPS, ALPHA, seed=3 + i, n_rounds=args.rounds,
                                                                k=TOPK, conf=conf)
            if record:
                record_stats(st, int((outdeg[srcs[i]] > 0).sum()))
            if world > 1:
                ids_blk.copy_(torch.from_numpy(ids))
                vals_blk.copy_(torch.from_numpy(vals))
        else:
            for j in range(q):
                s = int(srcs[i, j])
                _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + i, n_rounds=args.rounds, conf=conf, fetch=False)
                if record:
                    record_stats(st, int(outdeg[s] > 0))
                nsel, ids, vals, _, _ = g.topk_select(TOPK)
                if world > 1:
                    row_i = np.full(TOPK, -1, dtype=np.int32)
                    row_v = np.zeros(TOPK)
                    row_i[:len(ids)] = ids
                    row_v[:len(vals)] = vals
                    ids_blk[j].copy_(torch.from_numpy(row_i))
                    vals_blk[j].copy_(torch.from_numpy(row_v))
        if world > 1:  # the only exchange on the path: top-k blocks to rank 0 (xGMI / RCCL)
            dist.gather(ids_blk, gather_ids, dst=0)
            dist.gather(vals_blk, gather_vals, dst=0)

    for i in range(args.warmup):
        run_step(i, False)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.warmup, total_steps):
        run_step(i, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n_queries = args.steps * q * world
        value = n_queries / elapsed
        live_frac = acc["live"] / max(1, args.steps * q)
        # dominant kernel = the class with the largest summed HIP-event time over the timed region
        dom = max(range(1, 8), key=lambda c: acc["class_ms"][c])
        dom_ms, dom_bytes, dom_n = acc["class_ms"][dom], acc["class_bytes"][dom], acc["class_launches"][dom]
        achieved = (dom_bytes / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(pkg.KERNEL_NAMES[dom], {}).get("scale%d" % args.scale)
            except Exception:
                traffic = None
        roofline = {
            "bound": "hbm", "kernel": pkg.KERNEL_NAMES[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "launches": dom_n, "avg_launch_us": round(1e3 * dom_ms / max(1, dom_n), 2),
            "algorithmic_bytes_per_launch": int(dom_bytes / max(1, dom_n)),
            "other_kernels": {
                pkg.KERNEL_NAMES[c]: {
                    "ms": round(acc["class_ms"][c], 3), "launches": acc["class_launches"][c],
                    "achieved_GBps": round((acc["class_bytes"][c] / 1e9) / (acc["class_ms"][c] / 1e3), 1)
                    if acc["class_ms"][c] > 0 else 0.0}
                for c in (1, 2, 3, 5) if c != dom and acc["class_launches"][c]},
        }
        out = {
            "metric": "single-source PPR queries/sec (FORA, alpha=0.15, eps=0.5)", "value": round(value, 3),
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "RMAT scale-%d (n=%d, m=%d, seed 1) single-source FORA, %d uniformly drawn "
                                   "sources per step per GPU (seed 2; dead-end sources included)"
                                   % (args.scale, host.n, host.m, q),
                       "alpha": ALPHA, "eps": EPS, "queries_per_step": q, "rounds": args.rounds or "cost-model",
                       "mode": "16 queries in flight (pprhip_fora_batch_single_source), top-%d per query" % TOPK
                       if args.mode == "batch" else "one query at a time (pprhip_fora_single_source)",
                       "sharding": "replicated CSR, sources sharded by rank, top-%d gather to rank 0" % TOPK
                       if world > 1 else "single GPU"},
            "queries_per_s_live_sources": round(acc["live"] * world / elapsed, 3),
            "live_source_fraction": round(live_frac, 3),
            "avg_rounds": round(acc["rounds"] / max(1, args.steps * q), 2),
            "kernel_ms_per_live_query": {pkg.KERNEL_NAMES[c]: round(acc["class_ms"][c] / max(1, acc["live"]), 3)
                                         for c in (1, 2, 3, 5) if acc["class_launches"][c]},
            "dense_levels_per_live_query": round(acc["dense_levels"] / max(1, acc["live"]), 1),
            "levels_per_live_query": round(acc["levels"] / max(1, acc["live"]), 1),
            "useful_edge_fraction": round(acc["dense_edges"] / max(1, acc["dense_levels"] * host.m), 4),
            "graph_lift_s": {"generate_and_csr": round(t_gen, 2), "upload_and_tile": round(t_lift, 2)},
            "roofline": roofline,
        }
        if world == 1 and args.mode == "batch" and not args.no_extras:
            out["one_query_at_a_time"] = single_mode_sample(pkg, g, srcs[args.warmup], outdeg, conf, args)
        if world == 1 and args.mode == "batch" and not args.no_extras:
            out["topk_sample"] = topk_sample(pkg, g, srcs[args.warmup])
            out["all_pair_sample"] = all_pair_sample(pkg, g, host)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(host, srcs[args.warmup:], outdeg, live_frac, args.cpu_walk_divisor)
        print(json.dumps(out), flush=True)
    g.close()
    if world > 1:
        dist.destroy_process_group()


def single_mode_sample(pkg, g, srcs, outdeg, conf, args):
    """The same queries through pprhip_fora_single_source, one after another (latency path; outside the timed
    region): the first 32 sources of the first timed step, default cost-model profile."""
    g.set_tuning(pkg.tuning_default())
    sample = [int(s) for s in srcs[:32]]
    g.fora_single_source(sample[0], EPS, ALPHA, seed=1, n_rounds=args.rounds, conf=conf, fetch=False)
    ms, by, n_lv, live = 0.0, 0, 0, 0
    t0 = time.perf_counter()
    for j, s in enumerate(sample):
        _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + j, n_rounds=args.rounds, conf=conf, fetch=False)
        ms += st.class_ms[1]
        by += st.class_bytes[1]
        n_lv += st.class_launches[1]
        live += int(outdeg[s] > 0)
    dt = time.perf_counter() - t0
    return {"value": round(len(sample) / dt, 3), "unit": "queries/s", "queries": len(sample),
            "ms_per_live_query": round(1e3 * dt / max(1, live), 3),
            "dense_pull": {"launches": n_lv, "avg_launch_us": round(1e3 * ms / max(1, n_lv), 2),
                           "achieved_GBps": round((by / 1e9) / (ms / 1e3), 1) if ms > 0 else 0.0,
                           "frac": round((by / 1e9) / (ms / 1e3) / HBM_PEAK_GBS, 4) if ms > 0 else 0.0}}


def topk_sample(pkg, g, srcs):
    """FORA top-k (Fora_Topk, k = 32; configs #3 / #4) on the sources of the first timed step, outside the timed
    region: 16 queries in flight (pprhip_fora_batch_topk), and the first 32 of them one at a time."""
    g.set_tuning(pkg.tuning_default())
    srcs = np.ascontiguousarray(srcs, dtype=np.int32)
    g.fora_batch_topk(srcs[:16], TOPK, EPS, ALPHA, seed=1)
    t0 = time.perf_counter()
    ids, vals, st = g.fora_batch_topk(srcs, TOPK, EPS, ALPHA, seed=7)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    for j, s in enumerate(srcs[:32]):
        g.fora_topk(int(s), EPS, ALPHA, TOPK, seed=7 + j)
    dt1 = time.perf_counter() - t1
    return {"value": round(len(srcs) / dt, 1), "unit": "queries/s", "queries": int(len(srcs)), "k": TOPK,
            "rounds_per_query": round(st.rounds / max(1, len(srcs)), 2),
            "one_at_a_time_queries_per_s": round(32 / dt1, 1)}


def all_pair_sample(pkg, g, host):
    """The path's other workload, All-Pair-Backward-Search (config #5), on a bounded target range of the same
    graph (outside the timed region): 2^18 targets, threshold 1e-3, k = 32, index finalised on the host."""
    g.set_tuning(pkg.tuning_default())
    nt = min(host.n, 1 << 18)
    ix, _ = g.all_pair_backward(ALPHA, 1e-3, TOPK, 0, min(nt, 4096))
    ix.close()
    t0 = time.perf_counter()
    ix, st = g.all_pair_backward(ALPHA, 1e-3, TOPK, 0, nt)
    dt = time.perf_counter() - t0
    entries = int(len(ix.arrays()[1]))
    ix.close()
    return {"value": round(nt / dt, 1), "unit": "targets/s", "targets": nt, "threshold": 1e-3, "k": TOPK,
            "index_entries": entries, "tier2_targets": int(st.rounds), "tier3_targets": int(st.dense_nodes),
            "device_ms": round(st.total_ms, 1)}


def cpu_baseline(host, srcs, outdeg, live_frac, walk_divisor):
    """The CPU oracle's clock-driven FIFO FORA (the reference's algorithm as written, dense-array
    port) on one core: one live source of the timed batch, first push round in full, every
    `walk_divisor`-th walk, scaled back.  Dead-end sources cost ~0 on the CPU as well."""
    from oracle import oracle as orc
    og = orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)
    live = [int(s) for s in srcs.ravel() if outdeg[int(s)] > 0]
    if not live:
        return {"value": None, "unit": "queries/s", "cores": 1, "kind": "port", "sample": "no live source in batch"}
    s = live[0]
    t0 = time.time()
    _, push_s, walk_s, st = og.fora_whole_baseline(s, EPS, ALPHA, seed=3, walk_divisor=walk_divisor, max_rounds=0)
    wall = time.time() - t0
    t_live = push_s + walk_s * walk_divisor
    value = 1.0 / (t_live * max(live_frac, 1e-9))
    return {"value": round(value, 5), "unit": "queries/s", "cores": 1, "kind": "port",
            "sample": "source %d (out-degree %d): the reference's clock-driven push loop in full (%d FIFO rounds, "
                      "%.1f s, %d edge pushes) + every %d-th of its %d walks (%.1f s), scaled to one query = %.1f s; "
                      "divided by the batch's live-source fraction %.3f (dead-end sources return at once on the CPU "
                      "too); %.0f s of CPU work" % (s, int(outdeg[s]), st.rounds, push_s, st.edge_pushes,
                                                    walk_divisor, st.walks * walk_divisor, walk_s, t_live, live_frac,
                                                    wall),
            "seconds_per_live_query": round(t_live, 2)}


if __name__ == "__main__":
    main()
