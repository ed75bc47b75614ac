"""Multi-GPU sharding of the two batch workloads (one process per GPU, torch.distributed).

Queries and targets are independent (Gen_Util.java:208-232 loops queries, Base_Whole_Graph.java:76-92
loops targets), so the CSR is replicated and the work is sharded with no data-path collective:

  * batched FORA: query i runs on rank i mod world; the only exchange is a gather of the per-query
    top-k blocks to rank 0;
  * All-Pair-Backward-Search: rank r owns the contiguous target range target_range(r).  The result
    is keyed by *source* (Base_Whole_Graph.java:84-86), so one exchange follows: either every shard
    index is gathered to rank 0 and merged there (gather_index: small graphs), or - the form that
    scales - rank r also owns the sources target_range(r) and every rank sends each owner its rows
    (exchange_index_by_source: one all-to-all, a message per peer, i.e. per xGMI link), after which
    each rank merges the partial lists of its own sources with the reference's k rule
    (pprhip_index_merge) and holds that slice of the index.

Backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.  Only plain tensors travel.
"""
import numpy as np


def shard_sources(sources, rank, world):
    """Indices and ids of the queries rank `rank` runs: query i belongs to rank i mod world."""
    idx = np.arange(rank, len(sources), world)
    return idx, np.asarray(sources)[idx]


def target_range(rank, world, n):
    """Contiguous target range [begin, end) of rank `rank` for All-Pair-Backward-Search."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


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


def exchange_index_by_source(dist, torch, offsets, targets, values, rank, world, n, device="cpu"):
    """All-Pair's exchange by owner of the source.  `offsets/targets/values` is this rank's shard index
    (rows = all n sources, entries = its own targets).  Returns `world` partial indexes restricted to
    the sources this rank owns, as (offsets[n + 1], targets, values) ready for index_from_arrays +
    merge_indexes; rows outside the owned range are empty.  Rows arrive in target-shard order
    (sender 0 first), which is the reference's target-iteration order."""
    offsets = np.asarray(offsets, dtype=np.int64)
    targets = np.asarray(targets, dtype=np.int32)
    values = np.asarray(values, dtype=np.float64)
    ranges = [target_range(r, world, n) for r in range(world)]
    lo, hi = ranges[rank]
    lengths = np.diff(offsets)
    if world == 1:
        return [(offsets.astype(np.uint64), targets, values)]
    # 1) row lengths of every owner's range (split sizes are the range sizes, known everywhere)
    len_send = torch.as_tensor(lengths, dtype=torch.int64, device=device)
    len_recv = torch.empty((hi - lo) * world, dtype=torch.int64, device=device)
    dist.all_to_all_single(len_recv, len_send, output_split_sizes=[hi - lo] * world,
                           input_split_sizes=[e - b for b, e in ranges])
    # 2) the entries themselves: to owner o go the entries of rows [begin_o, end_o), contiguous in the CSR-style arrays
    cnt_send = [int(offsets[e] - offsets[b]) for b, e in ranges]
    len_recv_np = len_recv.cpu().numpy().reshape(world, hi - lo)
    cnt_recv = [int(x) for x in len_recv_np.sum(axis=1)]
    tg_recv = torch.empty(sum(cnt_recv), dtype=torch.int32, device=device)
    vl_recv = torch.empty(sum(cnt_recv), dtype=torch.float64, device=device)
    dist.all_to_all_single(tg_recv, torch.as_tensor(targets, device=device), output_split_sizes=cnt_recv,
                           input_split_sizes=cnt_send)
    dist.all_to_all_single(vl_recv, torch.as_tensor(values, device=device), output_split_sizes=cnt_recv,
                           input_split_sizes=cnt_send)
    tg_np, vl_np = tg_recv.cpu().numpy(), vl_recv.cpu().numpy()
    parts, at = [], 0
    for p in range(world):
        off = np.zeros(n + 1, dtype=np.uint64)
        off[lo + 1:hi + 1] = np.cumsum(len_recv_np[p])
        off[hi + 1:] = off[hi]
        parts.append((off, tg_np[at:at + cnt_recv[p]], vl_np[at:at + cnt_recv[p]]))
        at += cnt_recv[p]
    return parts
