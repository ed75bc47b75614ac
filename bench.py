#!/usr/bin/env python3
"""bench.py — single-source FORA queries/sec on a synthetic R-MAT graph (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

Workload (N = 1): R-MAT scale 22 (n = 4 194 304, m = 67 108 864, generator seed 1), single-source FORA with
alpha = 0.15, eps = 0.5.  A *step* is one batch of `--queries-per-step` sources drawn uniformly (seed 2) from the
nodes with out-degree > 0, handed to pprhip_fora_batch_single_source_resident: independent single-source
computations, 16 of them in flight, every query's whole-graph vector kept in a device-resident result store
(what getWholeGraphPPR() serves, Gen_Util.java:309) and its top-32 selected on the device.  Dead-end sources are
NOT drawn: they return at once in the reference (Forward_Push.java:72-76) and here, and counting them (the
reference's harness does, Gen_Util.java:99-107; 52 % of this graph's nodes) would inflate the rate 2.1x; the rate
with them in the mix is reported beside the headline as `value_uniform_sources`.  The graph is lifted into HBM
once before the timed region.  `value` = queries / second over all ranks.

Before the line is printed the results of the last timed step are checked on every rank: every vector's mass is
1 +- 1e-9 (device-side sum), every top-32 is sorted and consistent with the fetched vector.  A run that fails the
check prints no line.

N > 1 (one process per GPU, launched by torch.distributed.run): the CSR is replicated, every rank runs its own
batch per step (weak scaling, no data-path collective) and the per-step top-k blocks are gathered to rank 0 over
RCCL; `all_pair_scaling` reports the path's other workload, All-Pair-Backward-Search over all n targets, at the same
N (strong scaling; the exchange by owner of the source runs inside the library over RCCL; every rank's share runs in
a watched child process, so a collective that hangs or faults costs that sample, not the line).

Extra objects on the JSON line: `roofline` (dominant kernel class: HIP-event time on the engine's stream,
algorithmic bytes from DESIGN.md's byte model, HBM traffic from two `rocprofv3 --pmc` passes of this same build
run as child processes, useful-edge fraction of the sweeps) and `cpu_baseline` (the reference's algorithm on the
host cores: hash-map-shaped faithful port, dense-array port, all cores; rank 0 at N = 1 only); after the timed
region, at N = 1, also `one_query_at_a_time` (the drop-in path, pprhip_fora_single_source), `topk_sample`
(FORA top-32) and `all_pair_sample` (All-Pair-Backward-Search on 2^18 targets), each with its own `roofline`.
"""
import argparse
import csv
import glob
import importlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALPHA = 0.15
EPS = 0.5
TOPK = 32
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: 6.29 TB/s measured streaming copy (79 %)
RANDOM_REQUEST_ROOF_G = 55.0  # random requests beyond L2 per second on MI355X, measured: tools/micro/gather_rate.hip (8-byte
                              # gathers) and row_gather_rate.hip (rows up to 128 bytes: 52-55 G/s, flat from 128 MB to 2 GB
                              # tables).  Every such request moves a 128-byte line (profiles/r02_fetch_calibration.txt), so
                              # this roof is 6.7-7 TB/s of line traffic: the HBM roof met at line granularity.


def live_draw(rng, live_ids, shape):
    return live_ids[rng.integers(0, live_ids.size, size=shape)].astype(np.int32)


def self_check(pkg, store, srcs, ids, vals, nsel, per_query, n):
    """Mass and top-k consistency of the vectors of one step (raises SystemExit: no line for a wrong run)."""
    q = len(srcs)
    for i in range(q):
        s = store.sum(i)
        # floor(omega * rsum) = 0 walks leaves (1 - alpha) * residues undelivered, as in the reference
        ok = abs(s - 1.0) <= 1e-9 or (per_query[i].walks == 0 and abs(s + per_query[i].rsum - 1.0) <= 1e-9)
        if not ok:
            raise SystemExit("self-check failed: query %d (source %d) has mass %.12f" % (i, srcs[i], s))
        m = min(int(nsel[i]), TOPK)
        if m < 1 or np.any(np.diff(vals[i][:m]) > 0) or len(set(ids[i][:m].tolist())) != m or ids[i][:m].min() < 0 \
                or ids[i][:m].max() >= n:
            raise SystemExit("self-check failed: top-%d of query %d is not a sorted id list" % (TOPK, i))
    for i in sorted(set([0, q // 2, q - 1])):
        v = store.fetch(i)
        m = min(int(nsel[i]), TOPK)
        if not np.array_equal(v[ids[i][:m]], vals[i][:m]) or v.min() < 0.0:
            raise SystemExit("self-check failed: top-%d of query %d disagrees with its vector" % (TOPK, i))
        if m == TOPK and int((v >= vals[i][TOPK - 1]).sum()) != int(nsel[i]):
            raise SystemExit("self-check failed: entries >= the k-th value of query %d" % i)
    return {"queries": q, "mass_tolerance": 1e-9, "vectors_fetched": 3, "status": "ok"}


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
    ap.add_argument("--no-extras", action="store_true", help="skip the one-query-at-a-time, top-k and All-Pair samples")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 --pmc child passes (roofline.traffic)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--all-pair-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-walk-divisor", type=int, default=32)
    ap.add_argument("--tuning", default="", help="cost-model overrides, e.g. c_dense_edge_ns=0.002,max_rounds=30")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()
    if args.all_pair_child:
        return all_pair_child(args)

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
    live_ids = np.nonzero(outdeg > 0)[0]
    live_frac_graph = live_ids.size / host.n

    q = args.queries_per_step
    total_steps = args.warmup + args.steps
    rng = np.random.default_rng(2 + 7919 * rank)
    srcs = live_draw(rng, live_ids, (total_steps, q))
    store = pkg.Results(g, q) if args.mode == "batch" else None

    ids_blk = torch.empty((q, TOPK), dtype=torch.int32, device=xdev)
    vals_blk = torch.empty((q, TOPK), dtype=torch.float64, device=xdev)
    gather_ids = [torch.empty_like(ids_blk) for _ in range(world)] if (world > 1 and rank == 0) else None
    gather_vals = [torch.empty_like(vals_blk) for _ in range(world)] if (world > 1 and rank == 0) else None

    acc = {"class_ms": [0.0] * 8, "class_bytes": [0] * 8, "class_launches": [0] * 8, "rounds": 0, "queries": 0,
           "walks": 0, "walk_steps": 0, "levels": 0, "dense_levels": 0, "dense_edges": 0, "push_ms": 0.0, "mc_ms": 0.0}
    last = {}

    def record_stats(st, nq):
        for c in range(8):
            acc["class_ms"][c] += st.class_ms[c]
            acc["class_bytes"][c] += st.class_bytes[c]
            acc["class_launches"][c] += st.class_launches[c]
        acc["rounds"] += st.rounds
        acc["queries"] += nq
        acc["walks"] += st.walks
        acc["walk_steps"] += st.walk_steps
        acc["levels"] += st.levels
        acc["dense_levels"] += st.dense_levels
        acc["dense_edges"] += st.dense_edges
        acc["push_ms"] += st.push_ms
        acc["mc_ms"] += st.mc_ms

    def run_step(i, record):
        if args.mode == "batch":
            _, ids, vals, nsel, pq, st = g.fora_batch_single_source(srcs[i], EPS, ALPHA, seed=3 + i, n_rounds=args.rounds,
                                                                    k=TOPK, conf=conf, keep=store, per_query=True)
            last.update(step=i, ids=ids, vals=vals, nsel=nsel, pq=pq)
            if record:
                record_stats(st, q)
            if world > 1:
                ids_blk.copy_(torch.from_numpy(ids))
                vals_blk.copy_(torch.from_numpy(vals))
        else:
            for j in range(q):
                s = int(srcs[i, j])
                _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + i, n_rounds=args.rounds, conf=conf, fetch=False)
                if record:
                    record_stats(st, 1)
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

    check = None
    if args.mode == "batch" and last:
        check = self_check(pkg, store, srcs[last["step"]], last["ids"], last["vals"], last["nsel"], last["pq"], host.n)

    all_pair_scaling = None
    if not args.no_extras and not args.pmc_child and args.mode == "batch":
        all_pair_scaling = all_pair_scaling_sample(pkg, args, dist, torch, rank, world, local_rank, xdev)

    if rank == 0:
        n_queries = args.steps * q * world
        value = n_queries / elapsed
        nq = max(1, acc["queries"])
        # dominant kernel = the class with the largest summed HIP-event time over the timed region
        dom = max(range(1, 8), key=lambda c: acc["class_ms"][c])
        dom_ms, dom_bytes, dom_n = acc["class_ms"][dom], acc["class_bytes"][dom], acc["class_launches"][dom]
        achieved = (dom_bytes / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        avg_us = 1e3 * dom_ms / max(1, dom_n)
        useful = acc["dense_edges"] / max(1, acc["dense_levels"] * host.m)
        roofline = {
            "bound": "hbm", "kernel": pkg.KERNEL_NAMES[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
            "peak_achievable": HBM_ACHIEVABLE_GBS, "launches": dom_n, "avg_launch_us": round(avg_us, 2),
            "algorithmic_bytes_per_launch": int(dom_bytes / max(1, dom_n)),
            "useful_edge_fraction": round(useful, 4),
            "useful_note": "frontier edges of the levels run as sweeps / (sweeps x m): the share of a sweep's edge "
                           "gathers that carry a pushed residue; achieved x useful = %.0f GB/s of the roof spent on "
                           "real pushes" % (achieved * useful),
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
            "config": {"workload": "RMAT scale-%d (n=%d, m=%d, seed 1) single-source FORA, %d sources per step per GPU "
                                   "drawn uniformly (seed 2) from the nodes with out-degree > 0 (dead-end sources "
                                   "return at once and are not drawn)" % (args.scale, host.n, host.m, q),
                       "alpha": ALPHA, "eps": EPS, "queries_per_step": q, "rounds": args.rounds or "cost-model",
                       "mode": "16 queries in flight (pprhip_fora_batch_single_source_resident): every query's vector "
                               "kept in a device-resident store, top-%d per query" % TOPK
                       if args.mode == "batch" else "one query at a time (pprhip_fora_single_source)",
                       "sharding": "replicated CSR, sources sharded by rank, top-%d gather to rank 0" % TOPK
                       if world > 1 else "single GPU"},
            "ms_per_query": round(1e3 * elapsed / (args.steps * q), 3),
            "live_node_fraction_of_graph": round(live_frac_graph, 4),
            "value_vectors_resident": round(value, 3),
            "avg_rounds": round(acc["rounds"] / nq, 2),
            "kernel_ms_per_query": {pkg.KERNEL_NAMES[c]: round(acc["class_ms"][c] / nq, 3)
                                    for c in (1, 2, 3, 5) if acc["class_launches"][c]},
            "dense_levels_per_query": round(acc["dense_levels"] / nq, 1),
            "levels_per_query": round(acc["levels"] / nq, 1),
            "walks_per_query": int(acc["walks"] / nq),
            "graph_lift_s": {"generate_and_csr": round(t_gen, 2), "upload_and_tile": round(t_lift, 2)},
            "self_check": check,
            "roofline": roofline,
        }
        if all_pair_scaling is not None:
            out["all_pair_scaling"] = all_pair_scaling
        extras = world == 1 and args.mode == "batch" and not args.no_extras and not args.pmc_child
        if extras:
            out.update(delivery_samples(pkg, g, store, rng, live_ids, host, conf, q))
            out["one_query_at_a_time"] = single_mode_sample(pkg, g, srcs[args.warmup], conf, args, host)
            out["topk_sample"] = topk_sample(pkg, g, srcs[args.warmup])
            out["all_pair_sample"] = all_pair_sample(pkg, g, host)
        if args.pmc_child:  # the profiled child also runs the one-query-at-a-time path, for its kernels' counters
            single_mode_sample(pkg, g, srcs[args.warmup][:8], conf, args, host)
        if world == 1 and not args.no_pmc and not args.pmc_child and args.mode == "batch":
            pmc = pmc_traffic(args, host)
            roofline["traffic"] = pmc.get("dense_pull_batch")
            roofline["traffic_source"] = pmc["source"]
            if pmc.get("calibration") is not None:
                roofline["fetch_size_calibration"] = pmc["calibration"]
            if roofline["traffic"]:
                roofline["achieved_counter"] = round(roofline["traffic"] / 1e9 / (avg_us / 1e6), 1)
                roofline["frac_counter"] = round(roofline["achieved_counter"] / HBM_PEAK_GBS, 4)
                roofline["note"] = ("frac is SURVEY 8(d)'s algorithmic byte model over time and counts the gathers that "
                                    "L2 / LDS serve (a third of them), so it can exceed the HBM roof; frac_counter is "
                                    "the traffic the memory-side counters saw")
            if extras and pmc.get("dense_pull"):
                r1 = out["one_query_at_a_time"]["roofline"]
                r1["traffic"] = pmc["dense_pull"]
                r1["achieved_counter"] = round(pmc["dense_pull"] / 1e9 / (r1["avg_launch_us"] / 1e6), 1)
                r1["frac_counter"] = round(r1["achieved_counter"] / HBM_PEAK_GBS, 4)
            wk = roofline["other_kernels"].get("walk")
            if pmc.get("walk") and wk and wk.get("launches"):
                # the walk kernel's gathers use 4-8 bytes of every 128-byte line they move: bound by lines, not by
                # algorithmic bytes (writes are a twentieth of its traffic and counted with the lines here)
                wk["traffic_per_launch"] = pmc["walk"]
                avg_s = wk["ms"] / 1e3 / wk["launches"]
                wk["achieved_counter"] = round(pmc["walk"] / 1e9 / avg_s, 1)
                wk["frac_counter"] = round(wk["achieved_counter"] / HBM_PEAK_GBS, 4)
                wk["requests_beyond_l2_G_per_s"] = round(pmc["walk"] / 128.0 / avg_s / 1e9, 1)
                wk["random_request_roof_G_per_s"] = RANDOM_REQUEST_ROOF_G
                wk["frac_of_request_roof"] = round(wk["requests_beyond_l2_G_per_s"] / RANDOM_REQUEST_ROOF_G, 3)
        if world == 1 and not args.no_cpu_baseline and not args.pmc_child:
            out["cpu_baseline"] = cpu_baseline(host, srcs[args.warmup:], args.cpu_walk_divisor)
            if out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_faithful"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if store is not None:
        store.close()
    g.close()
    if world > 1:
        dist.destroy_process_group()


def delivery_samples(pkg, g, store, rng, live_ids, host, conf, q):
    """What reaches the caller, outside the timed region: the same batch with every vector fetched to the host
    (reserve_out: q x n doubles over PCIe), and with the reference's own source sampling (uniform over all nodes,
    dead ends included, Gen_Util.java:99-107)."""
    qf = q
    s = live_draw(rng, live_ids, qf)
    dest = np.zeros((qf, host.n))  # touched before the clock starts: the call is timed, not the kernel's page faults
    # (a first call with host delivery also pins the handle's staging buffers, as the headline's warm-up steps do theirs)
    g.fora_batch_single_source(s[:16], EPS, ALPHA, seed=11, k=TOPK, conf=conf, fetch=True, out=dest[:16])
    t0 = time.perf_counter()
    g.fora_batch_single_source(s, EPS, ALPHA, seed=11, k=TOPK, conf=conf, fetch=True, out=dest)
    dt_f = time.perf_counter() - t0
    del dest
    su = rng.integers(0, host.n, size=q).astype(np.int32)
    t0 = time.perf_counter()
    g.fora_batch_single_source(su, EPS, ALPHA, seed=12, k=TOPK, conf=conf, keep=store)
    dt_u = time.perf_counter() - t0
    return {"value_vectors_fetched": round(qf / dt_f, 3),
            "value_vectors_fetched_note": "%d queries, every whole-graph vector copied to pageable host memory inside "
                                          "the call (%.1f MB each); never the headline" % (qf, 8.0 * host.n / 1e6),
            "value_uniform_sources": round(q / dt_u, 3),
            "value_uniform_sources_note": "%d sources drawn uniformly from all nodes as Gen_Util.getQueryNodes does "
                                          "(%.1f %% of them dead ends that return at once)"
                                          % (q, 100.0 * float((np.diff(host.out_rp)[su] == 0).mean()))}


def single_mode_sample(pkg, g, srcs, conf, args, host):
    """The drop-in path: the same (live) sources through pprhip_fora_single_source, one after another, default
    cost-model profile; outside the timed region."""
    g.set_tuning(pkg.tuning_default())
    sample = [int(s) for s in srcs[:32]]
    g.fora_single_source(sample[0], EPS, ALPHA, seed=1, n_rounds=args.rounds, conf=conf, fetch=False)
    ms, by, n_lv, dl, de = 0.0, 0, 0, 0, 0
    cls = {1: 0.0, 2: 0.0, 3: 0.0}
    t0 = time.perf_counter()
    for j, s in enumerate(sample):
        _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + j, n_rounds=args.rounds, conf=conf, fetch=False)
        ms += st.class_ms[1]
        by += st.class_bytes[1]
        n_lv += st.class_launches[1]
        dl += st.dense_levels
        de += st.dense_edges
        for c in cls:
            cls[c] += st.class_ms[c]
    dt = time.perf_counter() - t0
    g.set_tuning(pkg.tuning_batch())
    ach = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    return {"value": round(len(sample) / dt, 3), "unit": "queries/s", "queries": len(sample),
            "ms_per_query": round(1e3 * dt / len(sample), 3),
            "kernel_ms_per_query": {pkg.KERNEL_NAMES[c]: round(v / len(sample), 3) for c, v in cls.items()},
            "dense_levels_per_query": round(dl / len(sample), 1),
            "roofline": {"bound": "hbm", "kernel": "dense_pull", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "launches": n_lv,
                         "avg_launch_us": round(1e3 * ms / max(1, n_lv), 2),
                         "algorithmic_bytes_per_launch": int(by / max(1, n_lv)),
                         "useful_edge_fraction": round(de / max(1, dl * host.m), 4)}}


def topk_sample(pkg, g, srcs):
    """FORA top-k (Fora_Topk, k = 32; configs #3 / #4) on the sources of the first timed step, outside the timed
    region: 16 queries in flight (pprhip_fora_batch_topk), and the first 32 of them one at a time.  Its roofline is
    the whole call's: algorithmic bytes of push (44 pops + 28 edges + 5 enqueues, sweeps 12m + 36n), walks and
    selections over the call's wall time."""
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
    g.set_tuning(pkg.tuning_batch())
    by = st.push_bytes + st.mc_bytes + st.select_bytes
    ach = by / 1e9 / dt
    return {"value": round(len(srcs) / dt, 1), "unit": "queries/s", "queries": int(len(srcs)), "k": TOPK,
            "rounds_per_query": round(st.rounds / max(1, len(srcs)), 2),
            "one_at_a_time_queries_per_s": round(32 / dt1, 1),
            "roofline": {"bound": "hbm", "kernel": "whole call (push + walks + selection)", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                         "algorithmic_bytes": int(by),
                         "class_ms": {pkg.KERNEL_NAMES[c]: round(st.class_ms[c], 2) for c in (2, 3, 5)
                                      if st.class_launches[c]}}}


def all_pair_sample(pkg, g, host):
    """The path's other workload, All-Pair-Backward-Search (config #5), on a bounded target range of the same
    graph (outside the timed region): 2^18 targets, threshold 1e-3, k = 32, index finalised on the host.  Roofline
    of its batched kernel: 44 B per pop + 28 B per edge + 16 B per index entry."""
    g.set_tuning(pkg.tuning_default())
    nt = min(host.n, 1 << 18)
    ix, _ = g.all_pair_backward(ALPHA, 1e-3, TOPK, 0, min(nt, 4096))
    ix.close()
    t0 = time.perf_counter()
    ix, st = g.all_pair_backward(ALPHA, 1e-3, TOPK, 0, nt)
    dt = time.perf_counter() - t0
    entries = int(len(ix.arrays()[1]))
    ix.close()
    g.set_tuning(pkg.tuning_batch())
    ms, by, nl = st.class_ms[4], st.class_bytes[4], st.class_launches[4]
    ach = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    return {"value": round(nt / dt, 1), "unit": "targets/s", "targets": nt, "threshold": 1e-3, "k": TOPK,
            "index_entries": entries, "tier2_targets": int(st.rounds), "tier3_targets": int(st.dense_nodes),
            "device_ms": round(st.total_ms, 1), "pops": int(st.pops), "edge_pushes": int(st.edge_pushes),
            "roofline": {"bound": "hbm", "kernel": "backward_batch (k_apbs)", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                         "launches": nl, "avg_launch_us": round(1e3 * ms / max(1, nl), 1),
                         "algorithmic_bytes_per_launch": int(by / max(1, nl)),
                         "note": "per-target state lives in LDS / HBM hash tables, one workgroup per target: the "
                                 "kernel is bound by chains of dependent probes and atomics (DESIGN.md 5), not by "
                                 "HBM streaming"}}


def _quiet_stdout(fn):
    """Runs fn with file descriptor 1 pointed at stderr: RCCL prints a version banner on stdout when it is first
    used, and this file's stdout is one JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        return fn()
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


def all_pair_scaling_sample(pkg, args, dist, torch, rank, world, local_rank, xdev):
    """All-Pair-Backward-Search over ALL n targets of the graph (config #5's shape; threshold 1e-3, k = 32) on `world`
    GPUs: strong scaling of the whole job.  Rank r searches the targets of its contiguous range, the entries are
    partitioned by owner of their source on the device and exchanged over RCCL inside the library
    (pprhip_all_pair_backward_sharded: one message per peer, one PCIe crossing per entry, at its owner), and every
    rank finalises the rows of its own sources.  Time = max over ranks.

    Every rank runs its share in a child process of its own (this file with --all-pair-child: own graph replica, own
    RCCL communicator from the id rank 0 made here): a collective that never returns or a fault inside it (a rank
    lost, a fabric error) then costs this sample, not the headline line - the child is killed at the limit and the
    line carries the error instead."""
    uid = [_quiet_stdout(pkg.comm_unique_id).hex() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(uid, src=0)
    limit = float(os.environ.get("PPRHIP_BENCH_WATCHDOG_S", "600"))
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), PPRHIP_COMM_ID=uid[0])
    cmd = [sys.executable, os.path.abspath(__file__), "--all-pair-child", "--scale", str(args.scale)]
    res, err = None, None
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        out, errtxt = child.communicate(timeout=limit)
        lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
        if child.returncode == 0 and lines:
            res = json.loads(lines[-1])
            if "error" in res:
                err, res = res["error"], None
        else:
            err = "child of rank %d exited with code %s: %s" % (rank, child.returncode, errtxt.decode(errors="replace")[-300:])
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        err = "no result from rank %d's child after %.0f s" % (rank, limit)
    # t_all, search seconds, entries found, bytes received, entries kept; a failed rank poisons the sample
    vals = [res["seconds"], res["search_seconds"], res["entries_found"], res["bytes_received"], res["entries_kept"]] \
        if res else [0.0] * 5
    stats = torch.tensor(vals + [0.0 if res else 1.0], dtype=torch.float64, device=xdev)
    tmax = stats.clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if float(stats[5]) > 0:
        return {"error": err or "%d of %d ranks failed" % (int(stats[5]), world)}
    n = 1 << args.scale
    return {"unit": "targets/s", "scaling": "strong", "targets": n, "threshold": 1e-3, "k": TOPK,
            "value": round(n / float(tmax[0]), 1), "seconds": round(float(tmax[0]), 3),
            "search_seconds_max_rank": round(float(tmax[1]), 3),
            "entries_found": int(stats[2]), "entries_kept_after_k_rule": int(stats[4]),
            "exchange_bytes_received": int(stats[3]),
            "exchange": "owner-of-source, 16-byte records partitioned on the device, grouped ncclSend/ncclRecv inside "
                        "libpprhip.so (pprhip_all_pair_backward_sharded); one child process per rank"}


def all_pair_child(args):
    """One rank's share of all_pair_scaling_sample, in a process of its own; prints one JSON object."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    try:
        import torch  # noqa: F401  first, as in the parent: the library then binds to the same HIP runtime and RCCL build
        pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, pkg.device_count())
        host = pkg.HostCsr.rmat(args.scale, 16, seed=1)
        with pkg.Graph(host, device=device) as g:
            comm = _quiet_stdout(lambda: pkg.Comm(g, bytes.fromhex(os.environ["PPRHIP_COMM_ID"]), rank, world))
            lo, hi = pkg.shard_target_range(rank, world, host.n)
            ix, _ = g.all_pair_backward(ALPHA, 1e-3, TOPK, lo, min(hi, lo + 1024))  # warm-up of the kernels
            ix.close()
            t0 = time.perf_counter()
            own, st = _quiet_stdout(lambda: comm.all_pair_backward_sharded(ALPHA, 1e-3, TOPK))
            t_all = time.perf_counter() - t0
            res = {"seconds": t_all, "search_seconds": st.total_ms / 1e3, "entries_found": float(st.mc_sources),
                   "bytes_received": float(st.select_bytes), "entries_kept": float(len(own.arrays()[1]))}
            own.close()
            comm.close()
    except Exception as e:  # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    print(json.dumps(res), flush=True)


# ---------------------------------------------------------------------------------------------- HBM counters
def _short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").strip()


def _pmc_pass(counter, args, workdir):
    """One `rocprofv3 --pmc <counter>` pass over a short child run of this same file (the program directly after
    `--`).  Returns {kernel: [values in KB]}."""
    d = os.path.join(workdir, counter)
    cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.abspath(__file__), "--pmc-child", "--steps", "1", "--warmup", "1", "--queries-per-step", "48",
           "--scale", str(args.scale), "--no-cpu-baseline", "--no-pmc"]
    if args.tuning:
        cmd += ["--tuning", args.tuning]
    env = dict(os.environ, TMPDIR="/tmp", PPRHIP_BATCH_THREADS="0")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode()[-300:]))
    out = {}
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == counter:
            out.setdefault(_short(row["Kernel_Name"]), []).append(float(row["Counter_Value"]))
    return out


def pmc_traffic(args, host):
    """HBM-side bytes per launch of the dominant kernel classes, measured on this build in this run: FETCH_SIZE and
    WRITE_SIZE in separate passes (they do not fit one pass on gfx950), KB units, and the guide's gfx950 correction:
    FETCH_SIZE = TCC_EA0_RDREQ x 64 B while every memory-side read request of this chip is 128 bytes - a coalesced
    stream and a random gather alike (profiles/r02_fetch_calibration.txt: TCC_EA0_RDREQ_128B = TCC_EA0_RDREQ for row
    gathers of 8 ... 512 bytes; FETCH_SIZE = 0.498 of the bytes of 128-byte rows; an 8-byte gather moves 128).  The
    factor is re-measured in every run on k_sum_partial, which reads exactly 8n bytes:
    bytes = FETCH_SIZE x 1024 x (8n / FETCH_SIZE(k_sum_partial)) + WRITE_SIZE x 1024."""
    if shutil.which("rocprofv3") is None:
        return {"source": "unmeasured: rocprofv3 not on PATH"}
    work = tempfile.mkdtemp(prefix="pprhip_pmc_", dir="/tmp")
    try:
        fetch = _pmc_pass("FETCH_SIZE", args, work)
        write = _pmc_pass("WRITE_SIZE", args, work)
    except Exception as e:  # the line is still valid without counters; say why they are missing
        shutil.rmtree(work, ignore_errors=True)
        return {"source": "unmeasured: %s" % str(e)[:200]}
    shutil.rmtree(work, ignore_errors=True)
    n = host.n

    def avg(d, k):
        v = d.get(k, [])
        return sum(v) / len(v) if v else 0.0

    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this build in this run; read bytes = "
                     "FETCH_SIZE x the factor measured on k_sum_partial (every memory-side read request is 128 bytes, "
                     "tallied at 64: MI355X_MICROARCH.md, profiles/r02_fetch_calibration.txt)"}
    factor = 2.0
    if fetch.get("k_sum_partial"):
        ratio = avg(fetch, "k_sum_partial") * 1024.0 / (8.0 * n)
        res["calibration"] = round(ratio, 3)
        if 0.4 < ratio < 1.1:
            factor = 1.0 / ratio

    def level_bytes(keys, levels):
        rd = sum(sum(fetch.get(k, [])) for k in keys) * 1024.0 * factor
        wr = sum(sum(write.get(k, [])) for k in keys) * 1024.0
        return int((rd + wr) / max(1.0, levels))

    # a dense level = several launches of the edge / apply kernels (one per Gauss-Seidel block) + one reduce launch
    bk = [k for k in fetch if k.startswith("k_dense_edges_b<") or k in ("k_dense_apply_batch", "k_dense_reduce_batch")]
    if bk and "k_dense_reduce_batch" in fetch:
        res["dense_pull_batch"] = level_bytes(bk, len(fetch["k_dense_reduce_batch"]))
    sk = [k for k in fetch if k.startswith("k_dense_edges<") or k.startswith("k_dense_apply<") or k == "k_dense_reduce"]
    if any(k.startswith("k_dense_edges<") for k in sk) and "k_dense_reduce" in fetch:
        # levels launched behind another one whose frontier had already emptied return at once and fetch next to
        # nothing: not counted (a counted level = one apply launch per Gauss-Seidel block, two blocks)
        levels = sum(sum(1 for x in v if x > 256.0) for k, v in fetch.items() if k.startswith("k_dense_apply<")) / 2.0
        res["dense_pull"] = level_bytes(sk, levels)
    if "k_mc_walk" in fetch:
        res["walk"] = int(avg(fetch, "k_mc_walk") * 1024.0 * factor + avg(write, "k_mc_walk") * 1024.0)
    return res


# ---------------------------------------------------------------------------------------------- CPU baselines
def cpu_baseline(host, srcs, walk_divisor):
    """The reference's CPU path on this box's host cores; the Java itself cannot run here.
    `value` is the faithful figure: the hash-map-shaped port, one thread, scaled from a bounded sample."""
    from oracle import baseline as base
    from oracle import oracle as orc
    og = orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)
    live = [int(s) for s in srcs.ravel()]
    cores = base.hardware_threads()
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    t_start = time.time()
    conf = og.conf_whole(ALPHA)
    _, omega = orc.fora_whole_params(conf, EPS)
    # (1) dense-array port, one core, one query in full (clock-driven loop), every walk_divisor-th walk
    a = base.fora_array_rounds(og, live[0], EPS, ALPHA, seed=3, walk_divisor=walk_divisor)
    a_push = sum(a["round_push_s"])
    a_walk_rate = a["walks_run"] / a["walk_s"] if a["walk_s"] > 0 else 0.0
    a_query = a_push + (a["walks_total"] / a_walk_rate if a_walk_rate else 0.0)
    # (2) hash-map-shaped faithful port, one core: bounded sample of the same source -> rates
    h = base.fora_hashmap(og, live[0], EPS, ALPHA, seed=3, walk_divisor=walk_divisor, push_budget_s=10.0)
    h_rate = h["edge_pushes"] / h["push_s"] if h["push_s"] > 0 else 0.0
    h_walk_rate = h["walks_run"] / h["walk_s"] if h["walk_s"] > 0 and h["walks_run"] else a_walk_rate
    # its clock-driven loop takes the turns whose push work the array port recorded, until push time exceeds
    # 400 ns x rsum x omega (Fora_Whole_Graph.java:93): slower pushes end the loop earlier, with more walks
    t_push, turns, rsum = 0.0, 0, conf.rsum
    for e, rs in zip(a["round_edge_pushes"], a["round_rsum"]):
        if not (t_push * 1e9 < 400.0 * rsum * omega):
            break
        t_push += e / h_rate if h_rate else 0.0
        rsum = rs
        turns += 1
    h_walks = omega * rsum
    h_query = t_push + (h_walks / h_walk_rate if h_walk_rate else 0.0)
    # (3) dense-array port over all cores, one query per thread
    par_srcs = live[1:1 + min(cores, 16)] or live[:1]
    p = base.fora_array_parallel(og, par_srcs, EPS, ALPHA, seed=3, walk_divisor=walk_divisor, threads=cores)
    # the run thinned the walks; per_query_s holds every query's time scaled to all of its walks, measured while the
    # other threads were running theirs: concurrent throughput = sum of the per-thread rates
    all_cores_qps = sum(1.0 / t for t in p["per_query_s"] if t > 0)
    wall = time.time() - t_start
    return {
        "value": round(1.0 / h_query, 6) if h_query > 0 else None, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": "faithful (hash-map-shaped) port of Forward_Push/Fora_Whole_Graph/Monte_Carlo on source %d: %.1f s "
                  "push sample (%d edge pushes, %.2f M/s; cut by the 10 s budget: %s) and %d walks (%.2f M/s); scaled "
                  "to one query with the per-turn push work of the array port's full run: %d turn(s) of the "
                  "clock-driven loop (%.1f s) + %.0f walks (%.1f s) = %.1f s per query; %.0f s of CPU work in all"
                  % (live[0], h["push_s"], h["edge_pushes"], h_rate / 1e6, h["truncated"], h["walks_run"],
                     h_walk_rate / 1e6, turns, t_push, h_walks, h_walks / h_walk_rate if h_walk_rate else 0.0,
                     h_query, wall),
        "seconds_per_query": round(h_query, 2),
        "faithful": {"edge_pushes_per_s": round(h_rate), "walks_per_s": round(h_walk_rate), "turns": turns,
                     "seconds_per_query": round(h_query, 2), "structures": "unordered_map<int64,double> x2, deque, "
                                                                           "unordered_set (HashMap/ConcurrentLinkedQueue/HashSet)"},
        "array": {"value": round(1.0 / a_query, 5) if a_query > 0 else None, "cores": 1,
                  "seconds_per_query": round(a_query, 2), "turns": a["rounds"], "push_s": round(a_push, 2),
                  "edge_pushes": int(sum(a["round_edge_pushes"])), "walks": int(a["walks_total"]),
                  "walks_per_s": round(a_walk_rate),
                  "sample": "source %d, clock-driven loop in full, every %d-th walk" % (live[0], walk_divisor)},
        "all_cores": {"value": round(all_cores_qps, 5), "cores": int(min(cores, len(par_srcs))),
                      "queries": len(par_srcs), "wall_s_thinned_walks": round(p["wall_s"], 2),
                      "seconds_per_query_each": [round(x, 1) for x in p["per_query_s"]],
                      "sample": "dense-array port, one query per thread on %d live sources at once; per-query times "
                                "scaled to all walks" % len(par_srcs)},
        "host": {"nproc": cores, "model": model},
    }


if __name__ == "__main__":
    main()
