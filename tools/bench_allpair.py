#!/usr/bin/env python3
"""All-Pair-Backward-Search throughput (config #5 shape), one process per GPU.

    python tools/bench_allpair.py --scale 22 --threshold 1e-3 --targets-per-rank 262144
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/bench_allpair.py ...

Weak scaling: every rank runs `--targets-per-rank` backward searches of its own contiguous target range on its
replica of the CSR, then the one exchange of the path follows (entries are keyed by source): each rank sends the
owner of a source that source's rows (all-to-all over RCCL) and merges what it receives with the reference's
k rule.  Prints one JSON line on rank 0: targets/s over all ranks, with and without the exchange + merge.
Not the headline metric (bench.py measures that); this is the tool behind DESIGN.md's All-Pair numbers."""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--threshold", type=float, default=1e-3)
    ap.add_argument("--k", type=int, default=32)
    ap.add_argument("--targets-per-rank", type=int, default=1 << 18)
    ap.add_argument("--backend", default="nccl")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # as bench.py runs
    import torch
    import torch.distributed as dist
    local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    sh = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd.sharding")
    host = pkg.HostCsr.rmat(args.scale, 16, seed=1)
    g = pkg.Graph(host, device=local_rank)
    lo, hi = sh.target_range(rank, world, host.n)
    hi = min(hi, lo + args.targets_per_rank)
    ix, _ = g.all_pair_backward(0.15, args.threshold, args.k, lo, min(hi, lo + 1024))  # warm-up
    ix.close()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ix, st = g.all_pair_backward(0.15, args.threshold, args.k, lo, hi)
    t_search = time.perf_counter() - t0
    off, tg, vl = ix.arrays()
    own = sh.exchange_index_by_source(dist, torch, off, tg, vl, rank, world, host.n, args.k, device=xdev)
    t_all = time.perf_counter() - t0
    times = torch.tensor([t_search, t_all], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(times, op=dist.ReduceOp.MAX)
    if rank == 0:
        n_t = (hi - lo) * world
        print(json.dumps({"metric": "All-Pair-Backward-Search targets/sec", "n_gpus": world, "scaling": "weak",
                          "targets": n_t, "threshold": args.threshold, "k": args.k,
                          "targets_per_s_search": round(n_t / float(times[0]), 1),
                          "targets_per_s_with_exchange_and_merge": round(n_t / float(times[1]), 1),
                          "entries_rank0_shard": int(len(tg)), "entries_rank0_owned": int(len(own.arrays()[1])),
                          "tier2_targets_rank0": st.rounds, "tier3_targets_rank0": int(st.dense_nodes),
                          "workload": "RMAT scale-%d (n=%d, m=%d)" % (args.scale, host.n, host.m)}), flush=True)
    ix.close()
    g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
