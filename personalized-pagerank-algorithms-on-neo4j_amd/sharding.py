"""Multi-GPU sharding of the two batch workloads (one process per GPU, torch.distributed).

Queries and targets are independent (Gen_Util.java:208-232 loops queries, Base_Whole_Graph.java:76-92
loops targets), so the CSR is replicated and the work is sharded with no data-path collective:

  * batched FORA: query i runs on rank i mod world; the only exchange is a gather of the per-query
    top-k blocks to rank 0;
  * All-Pair-Backward-Search: rank r owns the contiguous target range target_range(r).  The result
    is keyed by *source* (Base_Whole_Graph.java:84-86), so one exchange follows: either every shard
    index is gathered to rank 0 and merged there (gather_index: small graphs), or - the form that
    scales - rank r also owns the sources target_range(r) and every rank sends each owner its entries
    (exchange_index_by_source: partition by pprhip_owner_partition, one all-to-all per array, finalisation by
    pprhip_index_from_entries - the library's rule and the library's k rule, only the fabric is torch's), after
    which each rank holds the finished rows of its own sources.  The production exchange is the library's own
    (csrc/comm.cpp over RCCL); this module is the same protocol over any torch.distributed backend, which is what
    the CPU tests (gloo) and a rehearsal on fewer devices than ranks can run.

Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Only plain tensors travel.
"""
import numpy as np


def shard_sources(sources, rank, world):
    """Indices and ids of the queries rank `rank` runs: query i belongs to rank i mod world."""
    idx = np.arange(rank, len(sources), world)
    return idx, np.asarray(sources)[idx]


def target_range(rank, world, n):
    """Contiguous target range [begin, end) of rank `rank` for All-Pair-Backward-Search (= the sources it owns):
    the library's own rule (pprhip_shard_target_range)."""
    from . import shard_target_range
    return shard_target_range(rank, world, n)


def gather_topk(dist, torch, ids, vals, n_queries, k, rank, world, device="cpu"):
    """Gathers per-rank top-k rows (local query j = global query rank + j*world) to rank 0 and
    returns (ids[n_queries, k], vals[n_queries, k]) there, None elsewhere."""
    per_rank = (n_queries + world - 1) // world
    ids_t = torch.full((per_rank, k), -1, dtype=torch.int32, device=device)
    vals_t = torch.zeros((per_rank, k), dtype=torch.float64, device=device)
    if len(ids):
        ids_t[:len(ids)] = torch.as_tensor(np.asarray(ids, dtype=np.int32), device=device)
        vals_t[:len(vals)] = torch.as_tensor(np.asarray(vals, dtype=np.float64), device=device)
    if world == 1:
        return ids_t.cpu().numpy()[:n_queries], vals_t.cpu().numpy()[:n_queries]
    gi = [torch.empty_like(ids_t) for _ in range(world)] if rank == 0 else None
    gv = [torch.empty_like(vals_t) for _ in range(world)] if rank == 0 else None
    dist.gather(ids_t, gi, dst=0)
    dist.gather(vals_t, gv, dst=0)
    if rank != 0:
        return None
    out_i = np.full((n_queries, k), -1, dtype=np.int32)
    out_v = np.zeros((n_queries, k))
    for r in range(world):
        rows = np.arange(r, n_queries, world)
        out_i[rows] = gi[r].cpu().numpy()[:len(rows)]
        out_v[rows] = gv[r].cpu().numpy()[:len(rows)]
    return out_i, out_v


def gather_index(dist, torch, offsets, targets, values, rank, world, device="cpu"):
    """Gathers every rank's shard arrays to rank 0; returns a list of (offsets, targets, values)
    there, None elsewhere.  Entry counts differ per rank, so sizes travel first."""
    offsets = np.asarray(offsets, dtype=np.int64)
    if world == 1:
        return [(offsets.astype(np.uint64), np.asarray(targets), np.asarray(values))]
    cnt = torch.tensor([len(targets)], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cap = max(int(c.item()) for c in cnts)
    off_t = torch.as_tensor(offsets, device=device)
    tg_t = torch.zeros(cap, dtype=torch.int32, device=device)
    vl_t = torch.zeros(cap, dtype=torch.float64, device=device)
    tg_t[:len(targets)] = torch.as_tensor(np.asarray(targets, dtype=np.int32), device=device)
    vl_t[:len(values)] = torch.as_tensor(np.asarray(values, dtype=np.float64), device=device)
    go = [torch.empty_like(off_t) for _ in range(world)] if rank == 0 else None
    gt = [torch.empty_like(tg_t) for _ in range(world)] if rank == 0 else None
    gv = [torch.empty_like(vl_t) for _ in range(world)] if rank == 0 else None
    dist.gather(off_t, go, dst=0)
    dist.gather(tg_t, gt, dst=0)
    dist.gather(vl_t, gv, dst=0)
    if rank != 0:
        return None
    out = []
    for r in range(world):
        c = int(cnts[r].item())
        out.append((go[r].cpu().numpy().astype(np.uint64), gt[r].cpu().numpy()[:c], gv[r].cpu().numpy()[:c]))
    return out


def exchange_index_by_source(dist, torch, offsets, targets, values, rank, world, n, k, device="cpu"):
    """All-Pair's exchange by owner of the source over torch.distributed - the same three steps as the library's own
    RCCL exchange (csrc/comm.cpp: all_pair_sharded), with the library's own partition rule and finalisation:
    this rank's shard index is flattened to (source, target, value) entries, partitioned by owner of the source with
    pprhip_owner_partition (the function k_owner_partition evaluates on the device), every owner receives its entries
    in one all-to-all per array, and finalises them with pprhip_index_from_entries (bucketing by source, the
    reference's k rule).  Returns the finished Index of the sources this rank owns (rows of other sources empty)."""
    from . import index_from_entries, owner_partition
    offsets = np.asarray(offsets, dtype=np.int64)
    targets = np.asarray(targets, dtype=np.int32)
    values = np.asarray(values, dtype=np.float64)
    sources = np.repeat(np.arange(n, dtype=np.int32), np.diff(offsets))
    if world == 1:
        return index_from_entries(n, sources, targets, values, k)
    counts, order = owner_partition(n, world, sources)
    order = order.astype(np.int64)
    cnt_send = [int(c) for c in counts]
    # 1) sizes: one word per peer
    c_send = torch.as_tensor(np.asarray(cnt_send, dtype=np.int64), device=device)
    c_recv = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_to_all_single(c_recv, c_send)
    cnt_recv = [int(x) for x in c_recv.cpu().numpy()]
    # 2) the entries, owner by owner
    got = []
    for arr, dt in ((sources[order], torch.int32), (targets[order], torch.int32), (values[order], torch.float64)):
        recv = torch.empty(sum(cnt_recv), dtype=dt, device=device)
        dist.all_to_all_single(recv, torch.as_tensor(np.ascontiguousarray(arr), device=device),
                               output_split_sizes=cnt_recv, input_split_sizes=cnt_send)
        got.append(recv.cpu().numpy())
    # 3) the owner finalises its rows
    return index_from_entries(n, got[0], got[1], got[2], k)
