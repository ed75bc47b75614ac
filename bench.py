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
batch per step (weak scaling, no data-path collective) and the per-step top-k blocks are gathered on rank 0 by the
library's own entry point (pprhip_comm_create + pprhip_topk_gather: grouped ncclSend / ncclRecv inside libpprhip.so,
the exchange Gen_Util.java:208-232's loop needs when it is sharded); torch.distributed only carries the barrier and
the max-over-ranks of the clock.  The library's gather is probed once before the warm-up; should it fail on any
rank, all ranks agree to gather with torch.distributed instead and the line's `config.sharding` says why.  `all_pair_scaling` reports the path's other workload, All-Pair-Backward-Search over
all n targets, at the same N (strong scaling; the exchange by owner of the source runs inside the library over RCCL;
every rank's share runs in a watched child process, so a collective that hangs or faults costs that sample, not the
line).

Extra objects on the JSON line:
  `roofline`  dominant kernel class (HIP-event time on the engine's stream).  `achieved` / `frac`: the bytes the
              memory-side counters saw per launch over the launch's duration, against the 8 TB/s HBM peak - measured
              in this run by `rocprofv3 --pmc` child passes of this build (FETCH_SIZE, WRITE_SIZE, TCC hit / miss; one
              counter set per pass).  FETCH_SIZE counts requests that left L2, Infinity-Cache hits included, so this is
              L2-miss traffic: an upper bound of HBM traffic, below the peak by construction of the hardware.
              `frac_compulsory`: the same with every byte the sweep has to move counted once (a lower bound of its
              traffic: <= 1 by construction; traffic / compulsory = how often bytes are re-moved); `frac_model`:
              SURVEY 8(d)'s per-query gather model (counts gathers L2 / LDS serve; can exceed 1; kept for comparison
              with rounds 1-2).  Without counters (`--no-pmc`) `frac` falls back to the compulsory figure.
  `cpu_baseline`  the reference's algorithm on the host cores (rank 0, N = 1), in a background process: while the GPU
              measurements go on, the hash-map-shaped faithful port run to the end of two queries (one thread each,
              both times printed) and the dense-array port on two sources; after them, the array port on every core this job may use (its cgroup
              CPU quota), one query per core.
After the timed region, at N = 1: `one_query_at_a_time` (the drop-in path, pprhip_fora_single_source), `topk_sample`
(FORA top-32), `all_pair_sample` (All-Pair-Backward-Search on 2^18 targets) and `all_pair_rmat24` (config #5's graph
on one GPU, in a child process), each with its own `roofline` incl. counter traffic.
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

# Kernel arguments in device memory: read by the HIP runtime when it initialises, i.e. before torch's first HIP call
# (libpprhip.so sets the same default when it is loaded first, as in the `ppr` command line).  Inherited by
# the child processes; reported in the line as config.runtime_env.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PKG_NAME = "personalized-pagerank-algorithms-on-neo4j_amd"
HOOKS_LIB = os.path.join(ROOT, PKG_NAME, "libpprhip_hooks.so")  # the same sources with the test / measurement switches

ALPHA = 0.15
EPS = 0.5
TOPK = 32
AP_THR = 1e-3
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: 6.29 TB/s measured streaming copy (79 %)
RANDOM_REQUEST_ROOF_G = 55.0  # random 128-byte requests beyond L2 per second (tools/micro/gather_rate.hip)
RANDOM_ATOMIC_ROOF_G = 20.0   # random fp64 read-modify-writes per second, any form, 17-24 G/s by footprint
                              # (tools/micro/atomic_rate.hip, profiles/r03_atomic_rate.txt)


def load_host(pkg, scale):
    """The R-MAT host CSR (generator seed 1); cached in /tmp so that the child processes of one run (counter passes,
    All-Pair children, CPU baseline) do not generate it again."""
    path = "/tmp/pprhip_rmat%d_seed1.npz" % scale
    if os.path.exists(path):
        try:
            z = np.load(path)
            h = pkg.HostCsr.__new__(pkg.HostCsr)
            h.n, h.m = int(z["n"]), int(z["m"])
            h.out_rp, h.out_ci, h.in_rp, h.in_ci = z["out_rp"], z["out_ci"], z["in_rp"], z["in_ci"]
            if h.n == 1 << scale and h.out_rp.size == h.n + 1 and h.out_ci.size == h.m:
                return h
        except Exception:
            pass
    h = pkg.HostCsr.rmat(scale, 16, seed=1)
    if scale <= 24:  # (scale 24: 2.3 GB, read again by the counter passes over config #5's graph)
        try:
            tmp = path + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, n=h.n, m=h.m, out_rp=h.out_rp, out_ci=h.out_ci, in_rp=h.in_rp, in_ci=h.in_ci)
            os.replace(tmp, path)
        except Exception:
            pass
    return h


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
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (roofline traffic)")
    ap.add_argument("--no-rmat24", action="store_true", help="skip the R-MAT 24 All-Pair child sample")
    ap.add_argument("--rmat24-targets", type=int, default=1 << 24)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--trace-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--all-pair-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rmat24-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--topk-ab-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--rmat24-pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-sources", default="", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-walk-divisor", type=int, default=32)
    ap.add_argument("--tuning", default="", help="cost-model overrides, e.g. c_dense_edge_ns=0.002,max_rounds=30")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()
    if args.all_pair_child:
        return all_pair_child(args)
    if args.rmat24_child:
        return rmat24_child(args)
    if args.topk_ab_child:
        return topk_ab_child(args)
    if args.rmat24_pmc_child:
        return rmat24_pmc_child(args)
    if args.cpu_baseline_child:
        return cpu_baseline_child(args)
    if args.pmc_child:
        return pmc_child(args)
    if args.trace_child:
        return trace_child(args)

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
    xdev = "cuda" if args.backend == "nccl" else "cpu"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    # Kernel-class times are an option of the library since round 4 (event records between short kernels cost the
    # latency-bound paths 2-8 %).  The timed region runs WITH them - the roofline below needs the sweeps' durations over
    # the timed region, and the bandwidth-bound headline loses ~1 % - as do the All-Pair samples; the one-query-at-a-time
    # and top-k samples take their rates without them (as a caller gets them by default) and their class times in a
    # second pass.
    # Round 5: the timed region times the dense sweeps only (level 2: two records per sweep of 1.6 ms - the roofline's
    # class) and counts every other class; one more step outside the timed region runs with every class timed and
    # gives the per-class breakdown (kernel_ms_per_query, other_kernels).  The batch driver is one host thread that
    # feeds three streams: ~60 records per query on the small kernels' stream cost it 1.5 % (347-348 -> 353).
    # PPRHIP_BENCH_TIMING=0 / 1: developer A/B (no records at all / every class in the timed region, as until round 4).
    bench_timing = os.environ.get("PPRHIP_BENCH_TIMING", "2")
    pkg.set_kernel_timing({"0": False, "1": True}.get(bench_timing, 2))

    # ---- graph lift (outside the timed region)
    t0 = time.time()
    host = load_host(pkg, args.scale)
    t_gen = time.time() - t0
    outdeg = np.diff(host.out_rp)
    live_ids = np.nonzero(outdeg > 0)[0]
    live_frac_graph = live_ids.size / host.n
    q = args.queries_per_step
    total_steps = args.warmup + args.steps
    extra_step = 1 if args.mode == "batch" else 0  # (the step with every class timed, behind the timed region)
    rng = np.random.default_rng(2 + 7919 * rank)
    srcs = live_draw(rng, live_ids, (total_steps + extra_step, q))

    solo = world == 1 and rank == 0 and args.mode == "batch"
    t0 = time.time()
    g = pkg.Graph(host, device=local_rank)  # (before the children below: the lift's host half uses every core it may)
    t_lift = time.time() - t0
    # host-only work that runs beside the GPU measurements: the CPU baselines (a process of their own: they take
    # minutes of CPU time and use every core for a while) and the generation of config #5's graph
    cpu_child = None
    if solo and not args.no_cpu_baseline:
        cpu_child = start_cpu_baseline(args, srcs[args.warmup:].ravel())
    r24_child = None  # (started behind the timed region: its graph generation uses every core for a few seconds)
    if rank == 0:
        note("graph lifted (%.1f s generate / load, %.1f s lift)" % (t_gen, t_lift))
    conf = pkg.conf_whole_graph(host.n, host.m, ALPHA)
    tuning = pkg.tuning_batch() if args.mode == "batch" else pkg.tuning_default()
    for kv in filter(None, args.tuning.split(",")):
        key, val = kv.split("=")
        setattr(tuning, key, type(getattr(tuning, key))(float(val)))
    g.set_tuning(tuning)
    store = pkg.Results(g, q) if args.mode == "batch" else None

    # N > 1: the library's communicator for the top-k gather (rank 0 draws the id)
    comm = None
    comm_error = None
    if world > 1:
        uid = [_quiet_stdout(pkg.comm_unique_id).hex() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        if args.backend == "nccl":
            # The library's RCCL transport has never run with more than one rank before this line does (no multi-GPU
            # box was available to the build): every wait inside it is bounded (PPRHIP_COMM_TIMEOUT_S) and a rank's
            # failure reaches every peer, so a failure here costs the library gather, not the line - all ranks then
            # agree (over torch.distributed) to gather with torch.distributed instead, and the line says so.
            os.environ.setdefault("PPRHIP_COMM_TIMEOUT_S", "120")
            failed = 0
            try:
                comm = _quiet_stdout(lambda: pkg.Comm(g, bytes.fromhex(uid[0]), rank, world))
                probe = comm.topk_gather(np.full((1, TOPK), rank, dtype=np.int32), np.full((1, TOPK), float(rank)),
                                         rows_max=1)
                if rank == 0 and (probe is None or not all(int(probe[0][r][0][0]) == r for r in range(world))):
                    raise RuntimeError("the probe blocks did not come back rank by rank")
            except Exception as e:  # noqa: BLE001
                failed = 1
                comm_error = "%s: %s" % (type(e).__name__, str(e)[:200])
            flag = torch.tensor([failed], dtype=torch.int32, device=xdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if int(flag.item()):
                if comm is not None:  # the communicator may be broken: leave the group (abort), never a collective destroy
                    try:
                        comm.abort()
                        comm.close()
                    except Exception:  # noqa: BLE001
                        pass
                comm = None
                errs = [None] * world
                dist.all_gather_object(errs, comm_error)
                comm_error = "; ".join("rank %d: %s" % (r, e) for r, e in enumerate(errs) if e) or "a rank failed"

    if comm is not None:
        gather_how = "pprhip_topk_gather (RCCL inside libpprhip.so)"
    elif comm_error:
        gather_how = "torch.distributed.gather over %s, because the library's gather failed its probe (%s)" % (args.backend, comm_error)
    else:
        gather_how = ("torch.distributed.gather over %s (rehearsal on fewer devices than ranks: RCCL refuses two ranks on "
                      "one device)" % args.backend)
    acc = {"class_ms": [0.0] * 8, "class_bytes": [0] * 8, "class_launches": [0] * 8, "rounds": 0, "queries": 0,
           "walks": 0, "walk_steps": 0, "levels": 0, "dense_levels": 0, "dense_edges": 0, "push_ms": 0.0, "mc_ms": 0.0,
           "sweep_min_bytes": 0, "call_ms": 0.0, "walk_loads": 0, "walk_load_lanes": 0}
    last = {}
    gathered = {}

    def record_stats(st, nq):
        for c in range(8):
            acc["class_ms"][c] += st.class_ms[c]
            acc["class_bytes"][c] += st.class_bytes[c]
            acc["class_launches"][c] += st.class_launches[c]
        for k in ("rounds", "walks", "walk_steps", "levels", "dense_levels", "dense_edges", "push_ms", "mc_ms",
                  "sweep_min_bytes", "walk_loads", "walk_load_lanes"):
            acc[k] += getattr(st, k)
        acc["queries"] += nq
        acc["call_ms"] += st.total_ms

    def run_step(i, record):
        ids_blk = np.full((q, TOPK), -1, dtype=np.int32)
        vals_blk = np.zeros((q, TOPK))
        if args.mode == "batch":
            _, ids, vals, nsel, pq, st = g.fora_batch_single_source(srcs[i], EPS, ALPHA, seed=3 + i, n_rounds=args.rounds,
                                                                    k=TOPK, conf=conf, keep=store, per_query=True)
            last.update(step=i, ids=ids, vals=vals, nsel=nsel, pq=pq)
            if record:
                record_stats(st, q)
            ids_blk, vals_blk = ids, vals
        else:
            for j in range(q):
                s = int(srcs[i, j])
                _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + i, n_rounds=args.rounds, conf=conf, fetch=False)
                if record:
                    record_stats(st, 1)
                nsel, ids, vals, _, _ = g.topk_select(TOPK)
                ids_blk[j, :len(ids)] = ids
                vals_blk[j, :len(vals)] = vals
        if world > 1:  # the only exchange on the path: top-k blocks to rank 0
            if comm is not None:
                got = comm.topk_gather(ids_blk, vals_blk, rows_max=q)  # pprhip_topk_gather (RCCL inside the library)
                if got is not None:
                    gathered.update(ids=got[0], vals=got[1])
            else:  # gloo rehearsal on fewer devices than ranks: RCCL refuses two ranks on one device
                ti, tv = torch.from_numpy(ids_blk), torch.from_numpy(vals_blk)
                gl_i = [torch.empty_like(ti) for _ in range(world)] if rank == 0 else None
                gl_v = [torch.empty_like(tv) for _ in range(world)] if rank == 0 else None
                dist.gather(ti, gl_i, dst=0)
                dist.gather(tv, gl_v, dst=0)

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
        note("timed region done: %.1f queries/s" % (args.steps * q * world / elapsed))
    acc_timed = acc
    if extra_step and bench_timing == "2":
        # per-class times of the same workload: one more step, every class timed, outside the timed region
        acc = {k: (list(v) if isinstance(v, list) else v) for k, v in acc.items()}
        for k in acc:
            acc[k] = [type(x)(0) for x in acc[k]] if isinstance(acc[k], list) else type(acc[k])(0)
        was_t = pkg.set_kernel_timing(True)
        run_step(total_steps, True)  # (the self-check below then looks at this step's results: they are what the store holds)
        pkg.set_kernel_timing(was_t)
        acc_classes, acc = acc, acc_timed
    else:
        acc_classes = acc
    if solo and not args.no_extras and not args.no_rmat24:
        r24_child = start_rmat24(args)
    check = None
    if args.mode == "batch" and last:
        check = self_check(pkg, store, srcs[last["step"]], last["ids"], last["vals"], last["nsel"], last["pq"], host.n)
        if world > 1 and rank == 0 and gathered:  # rank 0's own block came back through the library's gather unchanged
            if not np.array_equal(gathered["ids"][0], last["ids"]) or not np.array_equal(gathered["vals"][0], last["vals"]):
                raise SystemExit("self-check failed: rank 0's top-k block changed in pprhip_topk_gather")
            check["gathered_blocks"] = int(gathered["ids"].shape[0])
    # N > 1: config #4 as BASELINE.json words it - 50 queries IN TOTAL over the N GPUs (strong scaling; SURVEY 8(e): GPU g
    # takes the sources i mod N = g), every rank under pprhip_tuning_batch_for(its share), the 50 x 32 pairs gathered on
    # rank 0 through the library when its communicator stands; beside the weak-scaling `value`, outside the timed region
    config4_strong = None
    if world > 1 and args.mode == "batch" and not args.no_extras:
        try:
            all50 = live_draw(np.random.default_rng(50), live_ids, 50)          # the same 50 sources on every rank
            mine = np.ascontiguousarray(all50[rank::world], dtype=np.int32)
            g.set_tuning(pkg.tuning_batch_for(len(mine)))
            was_t = pkg.set_kernel_timing(False)
            g.fora_batch_single_source(mine, EPS, ALPHA, seed=71, k=TOPK, conf=conf, keep=store)  # warm-up of this shape
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            _, ids50, vals50, _, _, _ = g.fora_batch_single_source(mine, EPS, ALPHA, seed=72, k=TOPK, conf=conf, keep=store)
            if comm is not None:
                comm.topk_gather(ids50, vals50, rows_max=(50 + world - 1) // world)
            torch.cuda.synchronize()
            dist.barrier()
            t50 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=xdev)
            dist.all_reduce(t50, op=dist.ReduceOp.MAX)
            pkg.set_kernel_timing(was_t)
            g.set_tuning(tuning)
            config4_strong = {"queries_total": 50, "gpus": world, "queries_per_gpu_max": int((50 + world - 1) // world),
                              "seconds": round(float(t50.item()), 4), "queries_per_s": round(50 / float(t50.item()), 1),
                              "scaling": "strong", "gather": "pprhip_topk_gather" if comm is not None else "none (no library communicator in this run)"}
        except Exception as e:  # noqa: BLE001  (a sample must not cost the line)
            config4_strong = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            g.set_tuning(tuning)
    if comm is not None:
        comm.close()

    all_pair_scaling = None
    if not args.no_extras and args.mode == "batch":
        all_pair_scaling = all_pair_scaling_sample(pkg, args, dist, torch, rank, world, local_rank, xdev)
        if rank == 0:
            note("All-Pair scaling sample done")

    if rank == 0:
        n_queries = args.steps * q * world
        value = n_queries / elapsed
        nq = max(1, acc["queries"])
        # dominant kernel = the class with the largest summed HIP-event time over the timed region
        dom = max(range(1, 8), key=lambda c: acc["class_ms"][c])
        dom_ms, dom_bytes, dom_n = acc["class_ms"][dom], acc["class_bytes"][dom], acc["class_launches"][dom]
        model = (dom_bytes / 1e9) / (dom_ms / 1e3) if dom_ms > 0 else 0.0
        avg_us = 1e3 * dom_ms / max(1, dom_n)
        useful = acc["dense_edges"] / max(1, acc["dense_levels"] * host.m)
        is_sweep = dom in (1, 5)
        comp_bytes = acc["sweep_min_bytes"] / max(1, dom_n) if is_sweep else dom_bytes / max(1, dom_n)
        comp = comp_bytes / 1e9 / (avg_us / 1e6) if avg_us > 0 else 0.0
        nq_c = max(1, acc_classes["queries"])
        kernel_ms = sum(acc_classes["class_ms"][c] for c in range(1, 8)) * (nq / nq_c)  # (scaled to the timed region's queries)
        roofline = {
            "bound": "hbm", "kernel": pkg.KERNEL_NAMES[dom], "achieved": round(comp, 1), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(comp / HBM_PEAK_GBS, 4), "frac_basis": "compulsory bytes (no counters in this run)",
            "frac_basis_short": "compulsory",
            "traffic": None, "peak_achievable": HBM_ACHIEVABLE_GBS,
            # one "launch" of the class dense_pull_batch is one SWEEP: per Gauss-Seidel block an edge kernel and an apply
            # kernel, then one reduce kernel (roofline.kernels names them as rocprofv3 does, with their own durations)
            ("sweeps" if dom == 5 else "launches"): dom_n,
            ("avg_sweep_us" if dom == 5 else "avg_launch_us"): round(avg_us, 2),
            "algorithmic_bytes_per_launch": int(comp_bytes),
            "algorithmic_note": "compulsory bytes: index stream, every gatherable contribution line once, row sums out "
                                "and in, next contributions, the busy queries' residues - each counted once "
                                "(pprhip_stats_t.sweep_min_bytes; DESIGN.md 6)",
            "achieved_compulsory": round(comp, 1), "frac_compulsory": round(comp / HBM_PEAK_GBS, 4),
            "model_bytes_per_launch": int(dom_bytes / max(1, dom_n)),
            "achieved_model": round(model, 1), "frac_model": round(model / HBM_PEAK_GBS, 4),
            "frac_model_note": "SURVEY 8(d)'s model: 4m + busy x (8m + 36 rows + 4) per sweep - one 8-byte gather per edge "
                               "and query, although a gathered line serves 16 queries and a third of the gathers hit L2 "
                               "/ LDS; can exceed 1, kept for comparison with rounds 1-2",
            "useful_edge_fraction": round(useful, 4),
            "useful_note": "frontier edges of the levels run as sweeps / (sweeps x m): the share of a sweep's edge "
                           "gathers that carry a pushed residue",
            "other_kernels": {
                pkg.KERNEL_NAMES[c]: {
                    "ms": round(acc_classes["class_ms"][c], 3), "launches": acc_classes["class_launches"][c],
                    "achieved_GBps": round((acc_classes["class_bytes"][c] / 1e9) / (acc_classes["class_ms"][c] / 1e3), 1)
                    if acc_classes["class_ms"][c] > 0 else 0.0}
                for c in (1, 2, 3, 5) if c != dom and acc_classes["class_launches"][c]},
            "other_kernels_note": "class times of one more step of the same workload with every class timed, outside the "
                                  "timed region (which times the sweeps only)" if acc_classes is not acc else None,
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
                       "runtime_env": {"HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG")},
                       "sharding": "replicated CSR, sources sharded by rank, top-%d blocks gathered on rank 0 by %s"
                                   % (TOPK, gather_how) if world > 1 else "single GPU"},
            "ms_per_query": round(1e3 * elapsed / (args.steps * q), 3),
            "live_node_fraction_of_graph": round(live_frac_graph, 4),
            "value_vectors_resident": round(value, 3),
            "avg_rounds": round(acc["rounds"] / nq, 2),
            "kernel_ms_per_query": {pkg.KERNEL_NAMES[c]: round(acc_classes["class_ms"][c] / nq_c, 3)
                                    for c in (1, 2, 3, 5, 6) if acc_classes["class_launches"][c]},
            "kernel_class_time": host_gap(1e3 * elapsed / (args.steps * q), kernel_ms / nq),
            "dense_levels_per_query": round(acc["dense_levels"] / nq, 1),
            "levels_per_query": round(acc["levels"] / nq, 1),
            "walks_per_query": int(acc["walks"] / nq),
            "walk_steps_G_per_s": round(acc_classes["walk_steps"] / (acc_classes["class_ms"][3] / 1e3) / 1e9, 2)
            if acc_classes["class_ms"][3] > 0 else None,
            "walk_lanes_per_load": round(acc["walk_load_lanes"] / max(1, acc["walk_loads"]), 2),
            "graph_lift_s": {"generate_and_csr": round(t_gen, 2), "upload_and_tile": round(t_lift, 2)},
            "self_check": check,
            "roofline": roofline,
        }
        if all_pair_scaling is not None:
            out["all_pair_scaling"] = all_pair_scaling
        if config4_strong is not None:
            out["config4_strong_scaling"] = config4_strong
        extras = solo and not args.no_extras
        if extras:
            out.update(q50_sample(pkg, g, store, rng, live_ids, conf, value))
            out["config4_share_rates"] = config4_share_sample(pkg, g, store, rng, live_ids, conf)
            note("config #4 share rates done")
            out.update(delivery_samples(pkg, g, store, rng, live_ids, host, conf, q))
            note("delivery samples done")
            out["one_query_at_a_time"] = single_mode_sample(pkg, g, srcs[args.warmup], conf, args, host)
            out["topk_sample"] = topk_sample(pkg, g, srcs[args.warmup])
            note("single-query and top-k samples done")
            out["all_pair_sample"] = all_pair_sample(pkg, g, host)
            note("All-Pair sample done")
        if store is not None:
            store.close()
            store = None
        g.close()  # HBM back before the children that lift graphs of their own
        g = None
        if r24_child is not None:
            out["all_pair_rmat24"] = finish_rmat24(r24_child)
            note("R-MAT 24 All-Pair child done")
        if solo and not args.no_pmc:
            apply_counters(out, pmc_traffic(args, host), avg_us, extras)
            note("counter passes done")
            if r24_child is not None and "error" not in out["all_pair_rmat24"]:
                rmat24_counters(out["all_pair_rmat24"], out["roofline"].get("fetch_size_calibration"))
                note("R-MAT 24 counter passes done")
            sweeps_alone(args, out["roofline"])
            note("sweeps-alone child done")
            if extras and "topk_sample" in out:
                out["topk_sample"]["same_box_ab"] = topk_same_box_ab(args)
                note("top-k A/B children done")
            idle = stream_idle(args)
            out["compute_stream_idle_frac"] = idle.get("compute_stream_idle_frac")
            if idle.get("sweep_kernels"):
                out["roofline"]["kernels"] = idle.pop("sweep_kernels")
            out["stream_occupancy"] = idle
            note("kernel-trace pass done")
        if cpu_child is not None:
            note("waiting for the CPU baseline child")
            out["cpu_baseline"] = finish_cpu_baseline(cpu_child)
            if out["cpu_baseline"].get("value"):
                out["speedup_vs_cpu_faithful"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out), flush=True)
    if store is not None:
        store.close()
    if g is not None:
        g.close()
    if world > 1:
        dist.destroy_process_group()


def host_gap(wall_ms, kernel_ms):
    """Wall time per query beside the summed HIP-event time of all kernel classes.  Since round 3 a query's walk phase
    runs on a side stream beside the other queries' sweeps, so the classes overlap and their sum says nothing about
    idle time; `compute_stream_idle_frac` (a kernel trace of the same workload, stream_idle) does."""
    return {"wall_ms_per_query": round(wall_ms, 3), "kernel_ms_per_query_all_classes": round(kernel_ms, 3),
            "class_time_over_wall": round(kernel_ms / wall_ms, 3) if wall_ms > 0 else None,
            "note": "classes overlap (walks beside sweeps): idle time is compute_stream_idle_frac / stream_occupancy"}


def q50_sample(pkg, g, store, rng, live_ids, conf, value_128, calls=6):
    """Config #4 as the reference runs it: 50 sources per call (PPR.java:179, Gen_Util.java:208-232 loops them), the
    calls one after another; outside the timed region.  50 = 3 x 16 + 2: the last queries of every call run with
    free slots beside them, which the 128-query steps of the headline dilute."""
    q = 50
    srcs = live_draw(rng, live_ids, (calls + 1, q))
    was = pkg.set_kernel_timing(False)  # (as a caller runs by default)
    try:
        return _q50_sample(pkg, g, store, rng, live_ids, conf, value_128, calls, q, srcs)
    finally:
        pkg.set_kernel_timing(was)


def _q50_sample(pkg, g, store, rng, live_ids, conf, value_128, calls, q, srcs):
    g.fora_batch_single_source(srcs[0], EPS, ALPHA, seed=21, k=TOPK, conf=conf, keep=store)
    t0 = time.perf_counter()
    for i in range(1, calls + 1):
        g.fora_batch_single_source(srcs[i], EPS, ALPHA, seed=21 + i, k=TOPK, conf=conf, keep=store)
    dt = time.perf_counter() - t0
    v = calls * q / dt
    out = {"value_q50": round(v, 3),
           "value_q50_note": "%d calls of 50 live sources each (config #4's call shape, PPR.java:179), %.1f ms per call; "
                             "%.3f of the rate of the %d-query steps" % (calls, 1e3 * dt / calls, v / value_128, 128)}
    # the same 50-query blocks through the query stream (pprhip_fora_stream_*): submitted as they come, no drain between
    # them; the top-k blocks of a block are checked against the synchronous call's
    try:
        _, ids_ref, vals_ref, _, _, _ = g.fora_batch_single_source(srcs[calls], EPS, ALPHA, seed=21 + calls, k=TOPK,
                                                                    conf=conf, keep=store)
        with pkg.QueryStream(g, EPS, ALPHA, k=TOPK, conf=conf) as qs:
            qs.wait(qs.submit(srcs[0], 21, keep=store))
            t0 = time.perf_counter()
            tickets = [qs.submit(srcs[i], 21 + i, keep=store) for i in range(1, calls + 1)]
            got = [qs.wait(tk) for tk in tickets]
            dt = time.perf_counter() - t0
        if not (np.array_equal(got[-1][0], ids_ref) and np.max(np.abs(got[-1][1] - vals_ref)) <= 1e-12):
            raise RuntimeError("a streamed block's top-k differs from the synchronous call's")
        vs = calls * q / dt
        # ... and the headline's own block size: 4 blocks of 128 through one stream
        big = live_draw(rng, live_ids, (5, 128))
        with pkg.QueryStream(g, EPS, ALPHA, k=TOPK, conf=conf) as qs:
            qs.wait(qs.submit(big[0], 41, keep=store))
            t0 = time.perf_counter()
            for tk in [qs.submit(big[i], 41 + i, keep=store) for i in range(1, 5)]:
                qs.wait(tk)
            out["value_stream_128"] = round(4 * 128 / (time.perf_counter() - t0), 3)
        out["value_q50_stream"] = round(vs, 3)
        out["value_q50_stream_note"] = ("the same %d blocks of 50 submitted to one query stream and waited for in order: "
                                        "%.3f of the rate of the %d-query steps; last block's top-%d ids identical to the "
                                        "synchronous call's, values to 1e-12" % (calls, vs / value_128, 128, TOPK))
    except Exception as e:  # noqa: BLE001
        out["value_q50_stream"] = None
        out["value_q50_stream_note"] = "failed: %s" % str(e)[:200]
    return out


def config4_share_sample(pkg, g, store, rng, live_ids, conf, calls=5):
    """Config #4 sharded over N GPUs gives every GPU a call of ceil(50 / N) queries (Gen_Util.java:208-232: 50 queries;
    SURVEY 8(e): GPU g takes the sources i mod N = g): calls of 25, 13 and 7 live sources on THIS GPU, under the plain batch
    profile and under pprhip_tuning_batch_for(q), which values a dense level by the columns a call of q can fill.  From the
    times, what N such GPUs would deliver for the 50 queries (the slowest share decides; the gather of 50 x 32 pairs is
    microseconds) - a prediction from one GPU, not a measurement of N."""
    was = pkg.set_kernel_timing(False)
    out = {"what": "calls of q live sources, %d calls each, vectors kept on the device, top-%d per query" % (calls, TOPK),
           "shares": {}}
    try:
        t50 = None
        for q in (50, 25, 13, 7):
            srcs = live_draw(rng, live_ids, (calls + 1, q))
            row = {}
            for name, tun in (("batch_profile", pkg.tuning_batch()), ("profile_for_q", pkg.tuning_batch_for(q))):
                g.set_tuning(tun)
                g.fora_batch_single_source(srcs[0], EPS, ALPHA, seed=61, k=TOPK, conf=conf, keep=store)
                t0 = time.perf_counter()
                for i in range(1, calls + 1):
                    g.fora_batch_single_source(srcs[i], EPS, ALPHA, seed=61 + i, k=TOPK, conf=conf, keep=store)
                dt = (time.perf_counter() - t0) / calls
                row[name] = {"queries_per_s": round(q / dt, 1), "ms_per_call": round(1e3 * dt, 2)}
            best = min(row["batch_profile"]["ms_per_call"], row["profile_for_q"]["ms_per_call"])
            if q == 50:
                t50 = best
            else:
                n_gpus = {25: 2, 13: 4, 7: 8}[q]
                row["predicted_strong_scaling"] = {"gpus": n_gpus, "queries_per_s": round(50 / (best / 1e3), 1),
                                                   "speedup_over_one_gpu": round(t50 / best, 2)}
            out["shares"]["q%d" % q] = row
    finally:
        g.set_tuning(pkg.tuning_batch())
        pkg.set_kernel_timing(was)
    return out


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


def single_mode_sample(pkg, g, srcs, conf, args, host, count=32):
    """The drop-in path: the same (live) sources through pprhip_fora_single_source, one after another, default
    cost-model profile; outside the timed region."""
    g.set_tuning(pkg.tuning_default())
    sample = [int(s) for s in srcs[:count]]
    g.fora_single_source(sample[0], EPS, ALPHA, seed=1, n_rounds=args.rounds, conf=conf, fetch=False)
    was = pkg.set_kernel_timing(False)  # the rate as a caller gets it by default ...
    t0 = time.perf_counter()
    for j, s in enumerate(sample):
        g.fora_single_source(s, EPS, ALPHA, seed=3 + j, n_rounds=args.rounds, conf=conf, fetch=False)
    dt_plain = time.perf_counter() - t0
    pkg.set_kernel_timing(True)  # ... and the class times in a second pass over the same queries
    ms, by, n_lv, dl, de, mb = 0.0, 0, 0, 0, 0, 0
    wsteps, wloads, wlanes = 0, 0, 0
    cls = {1: 0.0, 2: 0.0, 3: 0.0}
    t0 = time.perf_counter()
    for j, s in enumerate(sample):
        _, st = g.fora_single_source(s, EPS, ALPHA, seed=3 + j, n_rounds=args.rounds, conf=conf, fetch=False)
        ms += st.class_ms[1]
        by += st.class_bytes[1]
        n_lv += st.class_launches[1]
        dl += st.dense_levels
        de += st.dense_edges
        mb += st.sweep_min_bytes
        wsteps += st.walk_steps
        wloads += st.walk_loads
        wlanes += st.walk_load_lanes
        for c in cls:
            cls[c] += st.class_ms[c]
    dt_timed = time.perf_counter() - t0
    dt = dt_plain
    pkg.set_kernel_timing(was)
    g.set_tuning(pkg.tuning_batch())
    avg_s = ms / 1e3 / max(1, n_lv)
    model = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    comp = (mb / max(1, n_lv)) / 1e9 / avg_s if avg_s > 0 else 0.0
    return {"value": round(len(sample) / dt, 3), "unit": "queries/s", "queries": len(sample),
            "ms_per_query": round(1e3 * dt / len(sample), 3),
            "ms_per_query_with_kernel_timing": round(1e3 * dt_timed / len(sample), 3),
            "kernel_ms_per_query": {pkg.KERNEL_NAMES[c]: round(v / len(sample), 3) for c, v in cls.items()},
            "dense_levels_per_query": round(dl / len(sample), 1),
            "walk_steps_G_per_s": round(wsteps / (cls[3] / 1e3) / 1e9, 2) if cls[3] > 0 else None,
            "walk_lanes_per_load": round(wlanes / max(1, wloads), 2),
            "roofline": {"bound": "hbm", "kernel": "dense_pull", "achieved": round(comp, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(comp / HBM_PEAK_GBS, 4),
                         "frac_basis": "compulsory bytes (no counters in this run)", "traffic": None, "launches": n_lv,
                         "avg_launch_us": round(1e3 * ms / max(1, n_lv), 2),
                         "algorithmic_bytes_per_launch": int(mb / max(1, n_lv)),
                         "frac_compulsory": round(comp / HBM_PEAK_GBS, 4),
                         "model_bytes_per_launch": int(by / max(1, n_lv)), "frac_model": round(model / HBM_PEAK_GBS, 4),
                         "l2_hit_gather_roof_G_per_s": 250.0,
                         "gathers_G_per_s": round(host.m / avg_s / 1e9, 1) if avg_s > 0 else 0.0,
                         "useful_edge_fraction": round(de / max(1, dl * host.m), 4)}}


def topk_sample(pkg, g, srcs, count=None, single=32):
    """FORA top-k (Fora_Topk, k = 32; configs #3 / #4) on the sources of the first timed step, outside the timed
    region: 16 queries in flight (pprhip_fora_batch_topk), and the first 32 of them one at a time.  Its roofline is
    the whole call's: algorithmic bytes of push (44 pops + 28 edges + 5 enqueues, sweeps 12m + 36n), walks and
    selections over the call's wall time."""
    g.set_tuning(pkg.tuning_default())
    srcs = np.ascontiguousarray(srcs[:count] if count else srcs, dtype=np.int32)
    g.fora_batch_topk(srcs[:16], TOPK, EPS, ALPHA, seed=1)
    was = pkg.set_kernel_timing(False)  # the rates as a caller gets them by default
    t0 = time.perf_counter()
    g.fora_batch_topk(srcs, TOPK, EPS, ALPHA, seed=7)
    dt = time.perf_counter() - t0
    g.fora_topk(int(srcs[0]), EPS, ALPHA, TOPK, seed=1)  # warm-up: first use creates the handle's second stream
    t1 = time.perf_counter()
    for j, s in enumerate(srcs[:single]):
        g.fora_topk(int(s), EPS, ALPHA, TOPK, seed=7 + j)
    dt1 = time.perf_counter() - t1
    pkg.set_kernel_timing(True)  # class times and counters from a second, timed call
    t0 = time.perf_counter()
    ids, vals, st = g.fora_batch_topk(srcs, TOPK, EPS, ALPHA, seed=7)
    dt_timed = time.perf_counter() - t0
    pkg.set_kernel_timing(was)
    g.set_tuning(pkg.tuning_batch())
    by = st.push_bytes + st.mc_bytes + st.select_bytes
    ach = by / 1e9 / dt
    return {"value": round(len(srcs) / dt, 1), "unit": "queries/s", "queries": int(len(srcs)), "k": TOPK,
            "rounds_per_query": round(st.rounds / max(1, len(srcs)), 2),
            "one_at_a_time_queries_per_s": round(min(single, len(srcs)) / dt1, 1),
            "value_with_kernel_timing": round(len(srcs) / dt_timed, 1),
            "roofline": {"bound": "hbm", "kernel": "whole call (push + walks + selection)", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "frac_basis": "algorithmic bytes over wall time", "traffic": None,
                         "algorithmic_bytes": int(by), "algorithmic_bytes_per_query": int(by / max(1, len(srcs))),
                         "class_ms": {pkg.KERNEL_NAMES[c]: round(st.class_ms[c], 2) for c in (2, 3, 5)
                                      if st.class_launches[c]}}}


def topk_ab_child(args):
    """One leg of topk_sample.same_box_ab: FORA top-k (k = 32) one query at a time on 48 live sources, rate on stdout."""
    pkg = importlib.import_module(PKG_NAME)
    host = load_host(pkg, args.scale)
    live_ids = np.nonzero(np.diff(host.out_rp) > 0)[0].astype(np.int32)
    srcs = live_draw(np.random.default_rng(2), live_ids, 128)[:48]
    with pkg.Graph(host, device=0) as g:
        g.set_tuning(pkg.tuning_default())
        for s in srcs[:4]:
            g.fora_topk(int(s), EPS, ALPHA, TOPK, seed=1)
        best = 0.0
        for rep in range(2):
            t0 = time.perf_counter()
            for j, s in enumerate(srcs):
                g.fora_topk(int(s), EPS, ALPHA, TOPK, seed=7 + j)
            best = max(best, len(srcs) / (time.perf_counter() - t0))
    print(json.dumps({"queries_per_s": round(best, 1)}), flush=True)


def topk_same_box_ab(args):
    """round 5's one-pass round-start kernels against the two-pass ones of rounds 1-4
    (PPRHIP_TOPK_OLD_PASSES=1) on THIS box, back to back: two children on libpprhip_hooks.so (the switch is a measurement
    switch, which the product library does not read), identical but for the variable."""
    out = {}
    for name, extra in (("one_pass", {}), ("old_passes", {"PPRHIP_TOPK_OLD_PASSES": "1"})):
        env = dict(os.environ, PPRHIP_LIB_PATH=HOOKS_LIB, **extra)
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--topk-ab-child", "--scale", str(args.scale)],
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        res = _child_json(child, 150, "the top-k A/B child (%s)" % name)
        out[name] = res.get("queries_per_s", res)
    if all(isinstance(v, float) for v in out.values()):
        out["one_pass_over_old"] = round(out["one_pass"] / out["old_passes"], 3)
    out["what"] = "one query at a time, 48 live sources, best of two passes, libpprhip_hooks.so in child processes"
    return out


def index_column_check(off, tg, vl, targets, column_of, thr, k, tol=1e-12, slack=0.0, tie=1e-9, source_range=None):
    """Columns of an All-Pair index against backward searches of the same targets (Base_Whole_Graph.java:76-92 and the
    k rule of :112-163).  off / tg / vl: the index (CSR by source, rows value-descending); column_of(t) -> the dense
    reserve vector of a backward search from t at rmax = thr.  For every sampled target t:
      (1) every index entry (v, t, pi) is an entry of the search: |pi - column[v]| <= tol + slack, column[v] >= thr
          (less `tie` and slack), and no (v, t) pair occurs twice;
      (2) every entry the search yields with column[v] >= thr + tie + slack is in row v - unless the k rule cut it:
          row v then holds >= k entries, all of them >= that value (less tol + slack).
    slack = 0: the reference search runs the engine's schedule (values differ by summation order only);
    slack = thr: another push order (FIFO) - both reserves lie in [pi - thr, pi] (Backward_Search.java:89 leaves every
    residue <= thr).  source_range = (lo, hi): the index holds the rows of these sources only (one rank's share of a
    sharded run).  Returns counters; raises AssertionError with the first failures."""
    n = off.size - 1
    targets = np.unique(np.asarray(targets, dtype=np.int64))
    sel = np.zeros(n, dtype=bool)
    sel[targets] = True
    pos = np.nonzero(sel[tg])[0]
    rows = np.searchsorted(off, pos, side="right") - 1
    ptg = tg[pos].astype(np.int64)
    order = np.argsort(ptg, kind="stable")
    pos, rows, ptg = pos[order], rows[order], ptg[order]
    lo = np.searchsorted(ptg, targets, side="left")
    hi = np.searchsorted(ptg, targets, side="right")
    stats = {"targets": int(targets.size), "entries_checked": 0, "entries_cut_by_k_rule": 0, "max_abs_diff": 0.0}
    bad = []
    for j, t in enumerate(targets.tolist()):
        col = column_of(t)
        v_idx = rows[lo[j]:hi[j]]
        p_idx = vl[pos[lo[j]:hi[j]]]
        if np.unique(v_idx).size != v_idx.size:
            bad.append("target %d: a (source, target) pair occurs twice" % t)
        d = np.abs(p_idx - col[v_idx])
        if d.size:
            stats["max_abs_diff"] = max(stats["max_abs_diff"], float(d.max()))
        w = np.nonzero((d > tol + slack) | (col[v_idx] < thr - tie - slack))[0]
        for i in w[:3]:
            bad.append("target %d: index entry (source %d, %.17g) against the search's %.17g" % (t, v_idx[i], p_idx[i], col[v_idx[i]]))
        want = np.nonzero(col >= thr + tie + slack)[0]
        if source_range is not None:
            want = want[(want >= source_range[0]) & (want < source_range[1])]
        missing = np.setdiff1d(want, v_idx, assume_unique=False)
        if missing.size:
            cut = np.zeros(missing.size, dtype=bool)
            if k >= 1:  # (k < 1 keeps every entry: nothing may be missing)
                full = (off[missing + 1] - off[missing]).astype(np.int64) >= k
                if full.any():  # rows are value-descending: the last entry is the row's smallest
                    row_min = vl[(off[missing[full] + 1] - 1).astype(np.int64)]
                    cut[full] = row_min >= col[missing[full]] - tol - slack
            for v in missing[~cut][:3]:
                bad.append("target %d: entry (source %d, %.17g) of the search is not in the index and the k rule does not "
                           "explain it" % (t, v, col[v]))
            stats["entries_cut_by_k_rule"] += int(cut.sum())
        stats["entries_checked"] += int(v_idx.size)
        if len(bad) > 20:
            break
    if bad:
        raise AssertionError("All-Pair index disagrees with the backward searches of its targets:\n  " + "\n  ".join(bad[:20]))
    return stats


def all_pair_self_check(g, host, off, tg, vl, t_lo, t_hi, count=64, seed=5, source_range=None):
    """`count` columns of an index the bench has just built, against the engine's OWN single-target backward search
    (pprhip_backward_push: whole-vector levels in kernels_push.hip, not the All-Pair kernels; itself held to the
    oracle by tests/): the targets with the most in-edges of the range (shared levels, the full-size pass), the
    rest drawn at random.  A sample whose columns disagree raises: the caller prints the error, not a rate."""
    ind = np.diff(host.in_rp)[t_lo:t_hi]
    rng = np.random.default_rng(seed)
    hubs = (np.argsort(-ind.astype(np.int64), kind="stable")[:max(4, count // 8)] + t_lo).tolist()
    rest = (rng.integers(t_lo, t_hi, size=count - len(hubs))).tolist()

    def column_of(t):
        p, _, _ = g.backward_push(int(t), ALPHA, AP_THR)
        return p

    st = index_column_check(off, tg, vl, hubs + rest, column_of, AP_THR, TOPK, source_range=source_range)
    st.update(status="ok", against="pprhip_backward_push, 1e-12", hub_targets=len(hubs))
    return st


def all_pair_sample(pkg, g, host, nt=1 << 18):
    """The path's other workload, All-Pair-Backward-Search (config #5), on a bounded target range of the same
    graph (outside the timed region): 2^18 targets, threshold 1e-3, k = 32, index finalised on the host.  Roofline
    of its batched kernels: 44 B per pop + 28 B per edge + 16 B per index entry; the bound that
    binds them is the rate of random fp64 read-modify-writes (one per edge), not bytes."""
    g.set_tuning(pkg.tuning_default())
    nt = min(host.n, nt)
    ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, min(nt, 4096))
    ix.close()
    # every class timed around this call (the parent runs at level 2 - the sweeps only - since round 5, which left this
    # sample's kernel time, rate and roofline at zero in BENCH_r05.json)
    was = pkg.set_kernel_timing(True)
    t0 = time.perf_counter()
    ix, st = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, nt)
    dt = time.perf_counter() - t0
    pkg.set_kernel_timing(was)
    off, tg, vl = ix.arrays()
    ix.close()
    try:
        check = all_pair_self_check(g, host, off, tg, vl, 0, nt)
    except AssertionError as e:  # a wrong index prints its error, not a rate
        g.set_tuning(pkg.tuning_batch())
        return {"error": "self-check failed: %s" % str(e)[:600]}
    g.set_tuning(pkg.tuning_batch())
    res = all_pair_report(pkg, st, nt, dt, int(len(tg)))
    res["self_check"] = check
    return res


def all_pair_report(pkg, st, nt, dt, entries):
    ms, by, nl = st.class_ms[4], st.class_bytes[4], st.class_launches[4]
    if nl > 0 and not ms > 0:  # launches counted but not timed: a roofline of zeros would be a lie, say so instead
        return {"error": "backward_batch: %d launches with no kernel time (kernel timing was not on around the call)" % nl,
                "value": round(nt / dt, 1), "unit": "targets/s", "targets": nt, "seconds": round(dt, 3)}
    ach = (by / 1e9) / (ms / 1e3) if ms > 0 else 0.0
    return {"value": round(nt / dt, 1), "unit": "targets/s", "targets": nt, "threshold": AP_THR, "k": TOPK,
            "seconds": round(dt, 3), "index_entries": entries,
            "tier_census": {"lds_tier_targets": int(nt - st.rounds), "dense_tier_targets": int(st.rounds),
                            "whole_vector_tier_targets": int(st.dense_nodes)},
            "device_ms": round(st.total_ms, 1), "pops": int(st.pops), "edge_pushes": int(st.edge_pushes),
            "roofline": {"bound": "hbm", "kernel": "backward_batch (k_apbs_lds + k_apbs_dense)", "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                         "frac_basis": "algorithmic bytes (44 B/pop + 28 B/edge + 16 B/entry) over kernel time",
                         "traffic": None, "launches": nl, "kernel_ms": round(ms, 2),
                         "algorithmic_bytes": int(by),
                         "edges_G_per_s": round(st.edge_pushes / (ms / 1e3) / 1e9, 2) if ms > 0 else 0.0,
                         "random_rmw_roof_G_per_s": RANDOM_ATOMIC_ROOF_G,
                         "note": "an edge is one random fp64 read-modify-write (LDS in the first tier, a memory-side "
                                 "atomic on a dense per-workgroup vector in the second): bound by the chip's 17-24 G "
                                 "random read-modify-writes per second, not by bytes (DESIGN.md 5)"}}


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


def note(msg):
    """Progress line on stderr (stdout is the one JSON line; a long run that prints nothing looks hung)."""
    print("[bench %6.1f s] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def _child_json(child, limit, what):
    """One JSON object from a child process, or {'error': ...}; the child is killed at the limit."""
    try:
        out, errtxt = child.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        child.kill()
        child.communicate()
        return {"error": "no result from %s after %.0f s" % (what, limit)}
    lines = [l for l in out.decode(errors="replace").splitlines() if l.startswith("{")]
    if child.returncode == 0 and lines:
        try:
            return json.loads(lines[-1])
        except Exception as e:  # noqa: BLE001
            return {"error": "%s printed no JSON: %s" % (what, e)}
    return {"error": "%s exited with code %s: %s" % (what, child.returncode, errtxt.decode(errors="replace")[-300:])}


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
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank), PPRHIP_COMM_ID=uid[0],
               PPRHIP_COMM_TIMEOUT_S=str(int(limit * 0.8)))
    cmd = [sys.executable, os.path.abspath(__file__), "--all-pair-child", "--scale", str(args.scale)]
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    res = _child_json(child, limit, "rank %d's All-Pair child" % rank)
    err = res.get("error")
    # t_all, search seconds, entries found, bytes received, entries kept; a failed rank poisons the sample
    vals = [res["seconds"], res["search_seconds"], res["entries_found"], res["bytes_received"], res["entries_kept"],
            res["columns_checked"], res["entries_checked"], res["hbm_in_use_gb"]] if not err else [0.0] * 8
    stats = torch.tensor(vals[:5] + [1.0 if err else 0.0] + vals[5:], dtype=torch.float64, device=xdev)
    tmax = stats.clone()
    tmin = stats.clone()
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    if float(stats[5]) > 0:
        return {"error": err or "%d of %d ranks failed" % (int(stats[5]), world)}
    n = 1 << args.scale
    return {"unit": "targets/s", "scaling": "strong", "targets": n, "threshold": AP_THR, "k": TOPK,
            "value": round(n / float(tmax[0]), 1), "seconds": round(float(tmax[0]), 3),
            "search_seconds_max_rank": round(float(tmax[1]), 3),
            # imbalance between the ranks' shares (the targets are cut by modelled work when equal counts of ids would
            # leave one rank with more than 1.15 x the mean: pprhip_shard_target_cuts)
            "search_seconds_min_mean_max": [round(float(tmin[1]), 3), round(float(stats[1]) / world, 3),
                                            round(float(tmax[1]), 3)],
            "entries_found": int(stats[2]), "entries_kept_after_k_rule": int(stats[4]),
            "exchange_bytes_received": int(stats[3]),
            "self_check": {"status": "ok", "against": "pprhip_backward_push, 1e-12, on every rank's own rows",
                           "targets": int(stats[6]), "entries_checked": int(stats[7])},
            "hbm_in_use_per_rank_gb_max": round(float(tmax[8]), 2),
            "exchange": "owner-of-source, 16-byte records partitioned on the device, grouped ncclSend/ncclRecv inside "
                        "libpprhip.so (pprhip_all_pair_backward_sharded); one child process per rank"}


def all_pair_child(args):
    """One rank's share of all_pair_scaling_sample, in a process of its own; prints one JSON object."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    try:
        import torch  # noqa: F401  first, as in the parent: the library then binds to the same HIP runtime and RCCL build
        pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
        pkg.set_kernel_timing(True)
        device = int(os.environ.get("LOCAL_RANK", "0")) % max(1, pkg.device_count())
        host = load_host(pkg, args.scale)
        with pkg.Graph(host, device=device) as g:
            comm = _quiet_stdout(lambda: pkg.Comm(g, bytes.fromhex(os.environ["PPRHIP_COMM_ID"]), rank, world))
            lo, hi = pkg.shard_target_range(rank, world, host.n)
            ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, lo, min(hi, lo + 1024))  # warm-up of the kernels
            ix.close()
            t0 = time.perf_counter()
            own, st = _quiet_stdout(lambda: comm.all_pair_backward_sharded(ALPHA, AP_THR, TOPK))
            t_all = time.perf_counter() - t0
            off, tg, vl = own.arrays()
            own.close()
            comm.close()
            # this rank's rows (sources [lo, hi), targets of every rank) against single-target searches
            check = all_pair_self_check(g, host, off, tg, vl, 0, host.n, count=64, seed=5 + rank, source_range=(lo, hi))
            # HBM this rank holds at the end of its share (CSR replica, in-edge records, the dense tier's workspaces, the
            # whole-vector state; the record and exchange buffers have come and gone): what a rank of the sharded job needs
            free_b, total_b = g.device_memory()
            res = {"seconds": t_all, "search_seconds": st.total_ms / 1e3, "entries_found": float(st.mc_sources),
                   "bytes_received": float(st.select_bytes), "entries_kept": float(len(tg)),
                   "columns_checked": float(check["targets"]), "entries_checked": float(check["entries_checked"]),
                   "hbm_in_use_gb": (total_b - free_b) / 1e9}
    except Exception as e:  # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    print(json.dumps(res), flush=True)


# ---------------------------------------------------------------------------------------------- config #5's graph
def start_rmat24(args):
    """Config #5's graph (R-MAT 24: n = 16.7 M, m = 268 M) through All-Pair-Backward-Search on ONE GPU, in a child
    process (its 4 GB of CSR and the tier-2 workspaces sized for n = 2^24 come and go with it).  The child generates
    the graph on the host while the parent measures, and waits for a line on its stdin before it touches the GPU."""
    cmd = [sys.executable, os.path.abspath(__file__), "--rmat24-child", "--rmat24-targets", str(args.rmat24_targets)]
    return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def finish_rmat24(child):
    try:
        child.stdin.write(b"go\n")
        child.stdin.flush()
    except Exception:  # noqa: BLE001  (the child has died: _child_json reports how)
        pass
    return _child_json(child, float(os.environ.get("PPRHIP_BENCH_RMAT24_S", "420")), "the R-MAT 24 child")


def rmat24_child(args):
    try:
        pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
        pkg.set_kernel_timing(True)
        t0 = time.time()
        host = load_host(pkg, 24)
        t_gen = time.time() - t0
        sys.stdin.readline()  # the parent's GPU measurements are over
        import torch  # noqa: F401
        t0 = time.time()
        with pkg.Graph(host, device=0) as g:
            t_lift = time.time() - t0
            nt = min(host.n, args.rmat24_targets)
            ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, 4096)  # warm-up: workspaces of 2^24-node vectors
            ix.close()
            t0 = time.perf_counter()
            ix, st = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, nt)
            dt = time.perf_counter() - t0
            off, tg, vl = ix.arrays()
            ix.close()
            check = all_pair_self_check(g, host, off, tg, vl, 0, nt)  # raises: the line then carries the error
            res = all_pair_report(pkg, st, nt, dt, int(len(tg)))
            res["self_check"] = check
            free_b, total_b = g.device_memory()
            res["hbm_in_use_gb"] = round((total_b - free_b) / 1e9, 2)  # CSR + records + dense-tier workspaces of n = 2^24
            res["workload"] = "RMAT scale-24 (n=%d, m=%d, seed 1), All-Pair-Backward-Search on the first %d targets, " \
                              "threshold %g, k = %d, one GPU" % (host.n, host.m, nt, AP_THR, TOPK)
            res["graph_lift_s"] = {"generate_and_csr": round(t_gen, 1), "upload_and_tile": round(t_lift, 1)}
    except Exception as e:  # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    print(json.dumps(res), flush=True)


R24_PMC_TARGETS = 1 << 20


def rmat24_pmc_child(args):
    """What the counter passes over config #5's graph profile: the graph lifted, a marker, All-Pair on the first 2^20
    targets, a marker."""
    import torch  # noqa: F401
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    host = load_host(pkg, 24)
    with pkg.Graph(host, device=0) as g:
        ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, 4096)  # first-use work (workspaces, in-edge records)
        ix.close()
        g.random_walks(np.array([0], dtype=np.int32), np.array([0], dtype=np.uint64), ALPHA, seed=1)  # marker
        ix, st = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, R24_PMC_TARGETS)
        ix.close()
        g.random_walks(np.array([0], dtype=np.int32), np.array([0], dtype=np.uint64), ALPHA, seed=1)
        print(json.dumps({"pops": int(st.pops), "edge_pushes": int(st.edge_pushes), "push_bytes": int(st.push_bytes)}),
              flush=True)


def rmat24_counters(res, calibration):
    """Counter traffic of All-Pair's kernels on config #5's graph: two time-limited `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE) over rmat24_pmc_child, the first 2^20 targets, per algorithmic byte of the same targets, applied to
    the sample's algorithmic bytes.  calibration: FETCH_SIZE bytes per byte read, as measured by the main counter passes
    of this run on k_sum_partial (0.5 on gfx950: 128-byte requests tallied at 64)."""
    ro = res["roofline"]
    if shutil.which("rocprofv3") is None:
        ro["traffic_note"] = "unmeasured: rocprofv3 not on PATH"
        return
    work = tempfile.mkdtemp(prefix="pprhip_pmc24_", dir="/tmp")
    limit = float(os.environ.get("PPRHIP_BENCH_RMAT24_PMC_S", "240"))
    rows, child_out = {}, None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, counter)
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                   os.path.abspath(__file__), "--rmat24-pmc-child"]
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE, timeout=limit)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode()[-200:]))
            lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            child_out = json.loads(lines[-1]) if lines else child_out
            byid = {}
            for row in csv.DictReader(open(files[0])):
                byid.setdefault(int(row["Dispatch_Id"]), (_short(row["Kernel_Name"]), {}))[1][row["Counter_Name"]] = \
                    float(row["Counter_Value"])
            rows[counter] = _phases([byid[k] for k in sorted(byid)])
    except Exception as e:  # noqa: BLE001  (the sample stays valid without counters)
        ro["traffic_note"] = "unmeasured: %s" % str(e)[:200]
        shutil.rmtree(work, ignore_errors=True)
        return
    shutil.rmtree(work, ignore_errors=True)
    fetch, write = rows["FETCH_SIZE"], rows["WRITE_SIZE"]
    if len(fetch) < 3 or len(write) < 3 or not child_out:
        ro["traffic_note"] = "unmeasured: the counter rows do not hold the expected phase markers"
        return
    factor = 1.0 / calibration if calibration and 0.4 < calibration < 1.1 else 2.0
    ap = lambda k: k.startswith("k_apbs")  # noqa: E731
    kb = lambda ph, c: sum(v.get(c, 0.0) for name, v in ph if ap(name))  # noqa: E731
    dense = lambda ph, c: sum(v.get(c, 0.0) for name, v in ph if name == "k_apbs_dense")  # noqa: E731
    tr = kb(fetch[1], "FETCH_SIZE") * 1024.0 * factor + kb(write[1], "WRITE_SIZE") * 1024.0
    tr_dense = dense(fetch[1], "FETCH_SIZE") * 1024.0 * factor + dense(write[1], "WRITE_SIZE") * 1024.0
    per_alg = tr / max(1, child_out["push_bytes"])
    total = per_alg * ro["algorithmic_bytes"]
    ach = total / 1e9 / (ro["kernel_ms"] / 1e3) if ro["kernel_ms"] > 0 else 0.0
    ro.update(traffic=int(total),
              traffic_note="memory-side counters (FETCH_SIZE x %.2f + WRITE_SIZE) of k_apbs_* on the first 2^20 targets of "
                           "this graph in two rocprofv3 --pmc child passes: %.3g bytes for %.3g algorithmic bytes, applied "
                           "to this sample's algorithmic bytes" % (factor, tr, child_out["push_bytes"]),
              achieved_counter=round(ach, 1), frac_counter=round(ach / HBM_PEAK_GBS, 4),
              traffic_over_algorithmic=round(per_alg, 2),
              dense_tier_share_of_traffic=round(tr_dense / max(1.0, tr), 3))


# ---------------------------------------------------------------------------------------------- HBM counters
def _short(name):
    return name.split("(")[0].replace("void ", "").replace("pprhip::", "").strip()


MARK = "k_walk_batch"  # pprhip_random_walk_batch's kernel: launched by nothing else in the counted child


def pmc_child(args):
    """The program the counter passes profile: every measured path once, in short, with a marker kernel between the
    phases (pprhip_random_walk_batch of one walk) so that the per-dispatch counter rows can be attributed."""
    import torch  # noqa: F401
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    host = load_host(pkg, args.scale)
    live_ids = np.nonzero(np.diff(host.out_rp) > 0)[0]
    rng = np.random.default_rng(2)
    srcs = live_draw(rng, live_ids, (args.warmup + args.steps, args.queries_per_step))[args.warmup]
    g = pkg.Graph(host, device=0)
    conf = pkg.conf_whole_graph(host.n, host.m, ALPHA)
    tuning = pkg.tuning_batch()
    for kv in filter(None, args.tuning.split(",")):
        key, val = kv.split("=")
        setattr(tuning, key, type(getattr(tuning, key))(float(val)))
    g.set_tuning(tuning)
    store = pkg.Results(g, 48)

    def mark():
        g.random_walks(np.array([int(srcs[0])], dtype=np.int32), np.array([0], dtype=np.uint64), ALPHA, seed=1)

    # phase 0: warm-up (first-use allocations, LDS opt-ins) - not counted
    g.fora_batch_single_source(srcs[:16], EPS, ALPHA, seed=3, k=TOPK, conf=conf, keep=store)
    mark()  # 1: the headline path
    g.fora_batch_single_source(srcs[:48], EPS, ALPHA, seed=4, k=TOPK, conf=conf, keep=store)
    store.sum(0)  # k_sum_partial: the calibration kernel (reads exactly 8n bytes)
    mark()  # 2: one query at a time
    single_mode_sample(pkg, g, srcs, conf, args, host, count=8)
    mark()  # 3: top-k, 16 in flight
    g.set_tuning(pkg.tuning_default())
    g.fora_batch_topk(np.ascontiguousarray(srcs[:16]), TOPK, EPS, ALPHA, seed=7)
    mark()  # 4: top-k, one at a time
    for j, s in enumerate(srcs[:4]):
        g.fora_topk(int(s), EPS, ALPHA, TOPK, seed=7 + j)
    mark()  # 5: All-Pair, 2^16 targets (warm-up before it would be counted too: first-use work is part of this phase's
    ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, 256)  # kernels but not of the two kernels that are read out)
    ix.close()
    mark()  # 6
    ix, _ = g.all_pair_backward(ALPHA, AP_THR, TOPK, 0, 1 << 16)
    ix.close()
    mark()
    store.close()
    g.close()


def trace_child(args):
    """What the kernel-trace pass profiles: the headline path as the timed region runs it - warm-up, a marker kernel, two
    steps of --queries-per-step live sources through pprhip_fora_batch_single_source_resident, a marker."""
    import torch  # noqa: F401
    pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
    host = load_host(pkg, args.scale)
    live_ids = np.nonzero(np.diff(host.out_rp) > 0)[0]
    rng = np.random.default_rng(2)
    srcs = live_draw(rng, live_ids, (args.warmup + args.steps, args.queries_per_step))
    g = pkg.Graph(host, device=0)
    conf = pkg.conf_whole_graph(host.n, host.m, ALPHA)
    tuning = pkg.tuning_batch()
    for kv in filter(None, args.tuning.split(",")):
        key, val = kv.split("=")
        setattr(tuning, key, type(getattr(tuning, key))(float(val)))
    g.set_tuning(tuning)
    store = pkg.Results(g, args.queries_per_step)
    g.fora_batch_single_source(srcs[0], EPS, ALPHA, seed=3, k=TOPK, conf=conf, keep=store)
    g.random_walks(np.array([int(srcs[0][0])], dtype=np.int32), np.array([0], dtype=np.uint64), ALPHA, seed=1)
    for i in (args.warmup, args.warmup + 1):
        g.fora_batch_single_source(srcs[min(i, len(srcs) - 1)], EPS, ALPHA, seed=3 + i, k=TOPK, conf=conf, keep=store)
    g.random_walks(np.array([int(srcs[0][0])], dtype=np.int32), np.array([0], dtype=np.uint64), ALPHA, seed=1)
    store.close()
    g.close()


def stream_idle(args):
    """compute_stream_idle_frac of the headline path: 1 - (union of the kernel intervals on the compute stream / wall
    time), from a `rocprofv3 --kernel-trace` child pass over two steps of the timed region's workload (trace_child).
    The compute stream is the one the batched sweeps (k_dense_edges_b) run on; the walk phases run on a side stream
    beside it, whose busy share is reported as well."""
    if shutil.which("rocprofv3") is None:
        return {"source": "unmeasured: rocprofv3 not on PATH"}
    work = tempfile.mkdtemp(prefix="pprhip_trace_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", work, "--", sys.executable,
               os.path.abspath(__file__), "--trace-child", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--queries-per-step", str(args.queries_per_step), "--scale", str(args.scale)]
        if args.tuning:
            cmd += ["--tuning", args.tuning]
        r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, timeout=300)
        files = glob.glob(os.path.join(work, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"source": "unmeasured: rocprofv3 --kernel-trace failed (rc %d): %s" % (r.returncode, r.stderr.decode()[-200:])}
        rows = list(csv.DictReader(open(files[0])))
    except Exception as e:  # noqa: BLE001
        return {"source": "unmeasured: %s" % str(e)[:200]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    key = "Stream_Id" if rows and "Stream_Id" in rows[0] and len({x["Stream_Id"] for x in rows}) > 1 else "Queue_Id"
    rows.sort(key=lambda x: int(x["Start_Timestamp"]))
    marks = [i for i, x in enumerate(rows) if _short(x["Kernel_Name"]) == MARK]
    if len(marks) < 2:
        return {"source": "unmeasured: the trace does not hold the two marker kernels"}
    region = rows[marks[-2] + 1:marks[-1]]
    sweeps = [x for x in region if _short(x["Kernel_Name"]).startswith("k_dense_edges_b")]
    if not sweeps:
        return {"source": "unmeasured: no batched sweep in the traced region"}
    main = max({x[key] for x in sweeps}, key=lambda q: sum(1 for x in sweeps if x[key] == q))
    t0 = min(int(x["Start_Timestamp"]) for x in region)
    t1 = max(int(x["End_Timestamp"]) for x in region)

    def union(sel):
        iv = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in sel)
        busy, cur_lo, cur_hi = 0, None, None
        for lo, hi in iv:
            if cur_hi is None or lo > cur_hi:
                if cur_hi is not None:
                    busy += cur_hi - cur_lo
                cur_lo, cur_hi = lo, hi
            else:
                cur_hi = max(cur_hi, hi)
        return busy + (cur_hi - cur_lo if cur_hi is not None else 0)

    wall = max(1, t1 - t0)
    on_main = [x for x in region if x[key] == main]
    others = [x for x in region if x[key] != main]
    # where the compute stream idles: gaps between consecutive kernels on it, by (kernel before -> kernel after)
    gaps = {}
    prev_end, prev_name = None, None
    for x in on_main:
        lo, hi, name = int(x["Start_Timestamp"]), int(x["End_Timestamp"]), _short(x["Kernel_Name"]).split("<")[0]
        if prev_end is not None and lo > prev_end:
            g = gaps.setdefault("%s -> %s" % (prev_name, name), [0, 0])
            g[0] += 1
            g[1] += lo - prev_end
        if prev_end is None or hi > prev_end:
            prev_end, prev_name = hi, name
    top = sorted(gaps.items(), key=lambda kv: -kv[1][1])[:8]
    # the kernels one sweep of the batched path is made of, under the names rocprofv3 prints, with their own durations
    per = {}
    for x in region:
        name = _short(x["Kernel_Name"])
        if name.startswith(("k_dense_edges_b", "k_dense_apply_batch", "k_dense_reduce_batch")):
            e = per.setdefault(name, [0, 0])
            e[0] += 1
            e[1] += int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
    n_sweeps = max(1, sum(v[0] for k, v in per.items() if k.startswith("k_dense_reduce_batch")))
    sweep_kernels = {"sweeps_traced": n_sweeps,
                     "per_sweep_us": round(sum(v[1] for v in per.values()) / 1e3 / n_sweeps, 1),
                     "kernels": [{"name": k, "launches_per_sweep": round(v[0] / n_sweeps, 2),
                                  "avg_launch_us": round(v[1] / 1e3 / v[0], 1)} for k, v in sorted(per.items())]}
    return {"sweep_kernels": sweep_kernels,
            "source": "rocprofv3 --kernel-trace child pass over two steps of the headline workload; intervals grouped by %s" % key,
            "compute_stream_idle_frac": round(1.0 - union(on_main) / wall, 4),
            "any_stream_idle_frac": round(1.0 - union(region) / wall, 4),
            "side_streams_busy_frac": round(union(others) / wall, 4),
            "wall_ms": round(wall / 1e6, 2), "kernels_on_compute_stream": len(on_main), "kernels_on_side_streams": len(others),
            "largest_gaps_on_compute_stream": [{"between": k, "count": v[0], "ms": round(v[1] / 1e6, 2),
                                                "avg_us": round(v[1] / 1e3 / v[0], 1)} for k, v in top],
            "queries": 2 * args.queries_per_step}


def _pmc_pass(counters, args, workdir):
    """One `rocprofv3 --pmc <counters>` pass over pmc_child (the program directly after `--`).  Returns the rows in
    dispatch order: [(kernel, {counter: value})]."""
    d = os.path.join(workdir, counters.replace(" ", "_"))
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--output-format", "csv", "-d", d, "--", sys.executable,
           os.path.abspath(__file__), "--pmc-child", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--queries-per-step", str(args.queries_per_step), "--scale", str(args.scale)]
    if args.tuning:
        cmd += ["--tuning", args.tuning]
    env = dict(os.environ, TMPDIR="/tmp", PPRHIP_BATCH_THREADS="0")
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420)
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if r.returncode != 0 or not files:
        raise RuntimeError("rocprofv3 --pmc %s failed (rc %d): %s" % (counters, r.returncode, r.stderr.decode()[-300:]))
    rows = {}
    for row in csv.DictReader(open(files[0])):
        did = int(row["Dispatch_Id"])
        rows.setdefault(did, (_short(row["Kernel_Name"]), {}))[1][row["Counter_Name"]] = float(row["Counter_Value"])
    return [rows[k] for k in sorted(rows)]


def _phases(rows):
    """Splits the dispatch rows at the marker kernel: phases[i] = rows between marker i-1 and marker i."""
    out, cur = [], []
    for name, vals in rows:
        if name == MARK:
            out.append(cur)
            cur = []
        else:
            cur.append((name, vals))
    out.append(cur)
    return out


def pmc_traffic(args, host):
    """Memory-side bytes of every measured path, from counters collected in this run on this build: FETCH_SIZE,
    WRITE_SIZE and TCC_HIT / TCC_MISS in separate passes (the guide: one counter set per pass), KB units, and the
    guide's gfx950 correction: FETCH_SIZE = TCC_EA0_RDREQ x 64 B while every memory-side read request of this chip
    is 128 bytes - a coalesced stream and a random gather alike (profiles/r02_fetch_calibration.txt).  The factor is
    re-measured in every run on k_sum_partial, which reads exactly 8n bytes:
    bytes = FETCH_SIZE x 1024 x (8n / FETCH_SIZE(k_sum_partial)) + WRITE_SIZE x 1024.
    FETCH_SIZE counts what leaves L2, Infinity-Cache hits included: L2-miss traffic, an upper bound of HBM traffic."""
    if shutil.which("rocprofv3") is None:
        return {"source": "unmeasured: rocprofv3 not on PATH"}
    work = tempfile.mkdtemp(prefix="pprhip_pmc_", dir="/tmp")
    try:
        fetch = _phases(_pmc_pass("FETCH_SIZE", args, work))
        note("counter pass FETCH_SIZE done")
        write = _phases(_pmc_pass("WRITE_SIZE", args, work))
        note("counter pass WRITE_SIZE done")
    except Exception as e:  # the line is still valid without counters; say why they are missing
        shutil.rmtree(work, ignore_errors=True)
        return {"source": "unmeasured: %s" % str(e)[:200]}
    try:
        tcc = _phases(_pmc_pass("TCC_HIT_sum TCC_MISS_sum", args, work))
    except Exception as e:  # noqa: BLE001
        tcc = None
        tcc_err = str(e)[:160]
    shutil.rmtree(work, ignore_errors=True)
    n = host.n
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum child passes of this build in "
                     "this run; read bytes = FETCH_SIZE x the factor measured on k_sum_partial (every memory-side read "
                     "request is 128 bytes, tallied at 64: MI355X_MICROARCH.md, profiles/r02_fetch_calibration.txt); "
                     "FETCH_SIZE includes Infinity-Cache hits"}
    if len(fetch) < 8 or len(write) < 8:
        res["source"] = "unmeasured: the counter rows do not hold the expected phase markers"
        return res
    factor = 2.0
    cal = [v["FETCH_SIZE"] for name, v in fetch[1] if name == "k_sum_partial" and "FETCH_SIZE" in v]
    if cal:
        # the phase's last launch is store.sum(0) over all n values (sums inside the queries stop at the live range)
        ratio = cal[-1] * 1024.0 / (8.0 * n)
        res["calibration"] = round(ratio, 3)
        if 0.4 < ratio < 1.1:
            factor = 1.0 / ratio

    def kb(phase_rows, counter, pred):
        return sum(v.get(counter, 0.0) for name, v in phase_rows if pred(name))

    def count(phase_rows, pred):
        return sum(1 for name, v in phase_rows if pred(name))

    def traffic(ph, pred):
        return kb(fetch[ph], "FETCH_SIZE", pred) * 1024.0 * factor + kb(write[ph], "WRITE_SIZE", pred) * 1024.0

    def hit_rate(ph, pred):
        if tcc is None or len(tcc) <= ph:
            return None
        h, m = kb(tcc[ph], "TCC_HIT_sum", pred), kb(tcc[ph], "TCC_MISS_sum", pred)
        return round(h / (h + m), 4) if h + m > 0 else None

    anyk = lambda name: True  # noqa: E731
    # phase 1: the headline path; a dense level = several launches of the edge / apply kernels + one reduce launch
    # (the kernels of one sweep under rocprofv3's names: both are templates since round 5)
    sweep = lambda k: k.startswith(("k_dense_edges_b<", "k_dense_apply_batch", "k_dense_reduce_batch"))  # noqa: E731
    levels = count(fetch[1], lambda k: k == "k_dense_reduce_batch")
    if levels:
        res["dense_pull_batch"] = int(traffic(1, sweep) / levels)
        res["dense_pull_batch_tcc_hit"] = hit_rate(1, sweep)
        res["dense_pull_batch_edges_tcc_hit"] = hit_rate(1, lambda k: k.startswith("k_dense_edges_b<"))
    walks = count(fetch[1], lambda k: k == "k_mc_walk")
    if walks:
        res["walk"] = int(traffic(1, lambda k: k == "k_mc_walk") / walks)
    # every kernel of the phase (48 queries in one call + the calibration sum over n values)
    res["headline_bytes_per_query"] = int((traffic(1, anyk) - 8.0 * n) / 48)
    res["headline_bytes_by_class"] = {
        "dense_pull_batch": int(traffic(1, sweep) / 48), "walk": int(traffic(1, lambda k: k in ("k_mc_walk", "k_mc_plan<0>")) / 48),
        "sparse_push": int(traffic(1, lambda k: k.startswith("k_sparse")) / 48)}
    # phase 2: one query at a time; levels launched behind another one whose frontier had already emptied return at
    # once and fetch next to nothing: not counted (a counted level = one apply launch per Gauss-Seidel block, two blocks)
    one = lambda k: (k.startswith(("k_dense_edges<", "k_dense_edges_panel", "k_panel_fold", "k_dense_apply<"))  # noqa: E731
                     or k == "k_dense_reduce")
    lv = sum(1 for name, v in fetch[2] if name.startswith("k_dense_apply<") and v.get("FETCH_SIZE", 0.0) > 256.0) / 2.0
    if lv:
        res["dense_pull"] = int(traffic(2, one) / lv)
        res["dense_pull_tcc_hit"] = hit_rate(2, one)
    res["single_query_bytes"] = int(traffic(2, anyk) / 9)  # 8 timed queries + the sample's own warm-up query
    # phases 3 / 4: top-k, whole phase per query
    res["topk_batch_bytes_per_query"] = int(traffic(3, anyk) / 16)
    res["topk_single_bytes_per_query"] = int(traffic(4, anyk) / 4)
    res["topk_batch_tcc_hit"] = hit_rate(3, anyk)
    # phase 6: All-Pair on 2^16 targets
    ap = lambda k: k.startswith("k_apbs")  # noqa: E731
    res["all_pair_kernel_bytes_2p16"] = int(traffic(6, ap))
    res["all_pair_phase_bytes_2p16"] = int(traffic(6, anyk))
    res["all_pair_dense_bytes_2p16"] = int(traffic(6, lambda k: k == "k_apbs_dense"))
    res["all_pair_tcc_hit"] = hit_rate(6, ap)
    if tcc is None:
        res["tcc_note"] = "TCC_HIT / TCC_MISS pass failed: %s" % tcc_err
    return res


def sweeps_alone(args, roofline):
    """roofline.sweeps_alone: the same workload in a child process with everything in stream order (the driver of
    rounds 1-4: PPRHIP_BATCH_SLOTS_BESIDE=0, PPRHIP_BATCH_WALKS_BESIDE=0), where a sweep has the memory system to
    itself.  Since round 5 the timed region's sweeps share it with the walk phases and the sparse levels of the
    queries that hold no column, so `frac` (measured live, as the contract asks) is the sweep's share of a busier chip;
    this is what the kernels reach on their own, with the throughput that ordering gives."""
    if roofline.get("kernel") != "dense_pull_batch":
        return
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--scale", str(args.scale),
           "--queries-per-step", str(args.queries_per_step), "--no-cpu-baseline", "--no-pmc", "--no-extras", "--no-rmat24"]
    if args.tuning:
        cmd += ["--tuning", args.tuning]
    # (the two switches are measurement switches: read by libpprhip_hooks.so only - the same sources with the test hooks)
    env = dict(os.environ, PPRHIP_BATCH_SLOTS_BESIDE="0", PPRHIP_BATCH_WALKS_BESIDE="0",
               PPRHIP_LIB_PATH=HOOKS_LIB)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    res = _child_json(child, 240, "the sweeps-alone child")
    if "error" in res:
        roofline["sweeps_alone"] = res
        return
    us = res["roofline"].get("avg_sweep_us")
    per = roofline.get("traffic") or roofline["algorithmic_bytes_per_launch"]
    ach = per / 1e9 / (us / 1e6) if us else 0.0
    roofline["sweeps_alone"] = {
        "avg_sweep_us": us, "achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
        "frac_of_achievable": round(ach / HBM_ACHIEVABLE_GBS, 4),
        "bytes_per_sweep": "counter traffic" if roofline.get("traffic") else "compulsory",
        "value_in_stream_order": res["value"],
        "note": "child run with every kernel of the job on one stream (no walk phase or sparse level beside a sweep): "
                "the sweep kernels' own rate, and the queries/s that ordering gives; `frac` above is measured live in the "
                "timed region, where the sweeps share the memory system with that other work"}


def apply_counters(out, pmc, avg_us, extras):
    """Writes the counter figures into the line's roofline objects."""
    roofline = out["roofline"]
    roofline["traffic_source"] = pmc["source"]
    if pmc.get("calibration") is not None:
        roofline["fetch_size_calibration"] = pmc["calibration"]
    tr = pmc.get("dense_pull_batch") if roofline["kernel"] == "dense_pull_batch" else None
    if tr:
        ach = tr / 1e9 / (avg_us / 1e6)
        roofline.update(traffic=tr, achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                        frac_of_achievable=round(ach / HBM_ACHIEVABLE_GBS, 4), frac_basis_short="counter",
                        frac_basis="memory-side counters (FETCH_SIZE x calibration + WRITE_SIZE) per launch over the "
                                   "launch's duration; FETCH_SIZE includes Infinity-Cache hits: L2-miss traffic",
                        achieved_counter=round(ach, 1), frac_counter=round(ach / HBM_PEAK_GBS, 4),
                        traffic_over_compulsory=round(tr / max(1, roofline["algorithmic_bytes_per_launch"]), 2),
                        tcc_hit_rate=pmc.get("dense_pull_batch_tcc_hit"),
                        tcc_hit_rate_edge_kernel=pmc.get("dense_pull_batch_edges_tcc_hit"))
    if pmc.get("headline_bytes_per_query"):
        # the whole job, every kernel class: what the two-handles experiment says is the binding resource
        # of the batched path - a second group of 16 queries beside the first adds nothing (325 against 324 queries/s)
        bq = pmc["headline_bytes_per_query"]
        ach = bq * out["value"] / 1e9
        roofline["whole_job"] = {
            "traffic_per_query": bq, "by_class": pmc.get("headline_bytes_by_class"),
            "achieved": round(ach, 1), "unit": "GB/s", "frac_of_peak": round(ach / HBM_PEAK_GBS, 4),
            "frac_of_achievable": round(ach / HBM_ACHIEVABLE_GBS, 4),
            "note": "memory-side bytes of every kernel of a 48-query call (counter passes) x the timed region's queries/s: "
                    "the rate at which the whole workload - sweeps, walks beside them, sparse levels, selections - moves "
                    "lines beyond L2, against the 8 TB/s peak and the streaming rate this chip reaches (%d GB/s)"
                    % int(HBM_ACHIEVABLE_GBS)}
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
    if not extras:
        return
    r1 = out["one_query_at_a_time"]["roofline"]
    if pmc.get("dense_pull"):
        ach = pmc["dense_pull"] / 1e9 / (r1["avg_launch_us"] / 1e6)
        r1.update(traffic=pmc["dense_pull"], achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBS, 4),
                  frac_basis="memory-side counters per level over the level's duration (FETCH_SIZE includes "
                             "Infinity-Cache hits)", frac_counter=round(ach / HBM_PEAK_GBS, 4),
                  traffic_over_compulsory=round(pmc["dense_pull"] / max(1, r1["algorithmic_bytes_per_launch"]), 2),
                  tcc_hit_rate=pmc.get("dense_pull_tcc_hit"), traffic_per_query=pmc.get("single_query_bytes"))
    rt = out["topk_sample"]["roofline"]
    if pmc.get("topk_batch_bytes_per_query"):
        qps = out["topk_sample"]["value"]
        ach = pmc["topk_batch_bytes_per_query"] * qps / 1e9
        rt.update(traffic=pmc["topk_batch_bytes_per_query"], traffic_unit="bytes per query, 16 in flight (whole call)",
                  achieved_counter=round(ach, 1), frac_counter=round(ach / HBM_PEAK_GBS, 4),
                  traffic_over_algorithmic=round(pmc["topk_batch_bytes_per_query"] / max(1, rt["algorithmic_bytes_per_query"]), 2),
                  traffic_per_query_one_at_a_time=pmc.get("topk_single_bytes_per_query"),
                  tcc_hit_rate=pmc.get("topk_batch_tcc_hit"))
    ra = out["all_pair_sample"]["roofline"]
    if pmc.get("all_pair_kernel_bytes_2p16"):
        scale = out["all_pair_sample"]["targets"] / float(1 << 16)
        tr = pmc["all_pair_kernel_bytes_2p16"] * scale
        ach = tr / 1e9 / (ra["kernel_ms"] / 1e3) if ra["kernel_ms"] > 0 else 0.0
        ra.update(traffic=int(tr), traffic_note="counted on the first 2^16 targets of the same range in the counter "
                                                "passes, scaled to this sample's targets",
                  achieved_counter=round(ach, 1), frac_counter=round(ach / HBM_PEAK_GBS, 4),
                  traffic_over_algorithmic=round(tr / max(1, ra["algorithmic_bytes"]), 2),
                  dense_tier_share_of_traffic=round(pmc.get("all_pair_dense_bytes_2p16", 0) / max(1, pmc["all_pair_kernel_bytes_2p16"]), 3),
                  tcc_hit_rate=pmc.get("all_pair_tcc_hit"))


# ---------------------------------------------------------------------------------------------- CPU baselines
def start_cpu_baseline(args, live):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--scale", str(args.scale),
           "--cpu-walk-divisor", str(args.cpu_walk_divisor), "--cpu-sources", ",".join(str(int(s)) for s in live[:300])]
    return subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def finish_cpu_baseline(child):
    # the GPU measurements are over: the child may now take every core this job has (its last measurement)
    try:
        child.stdin.write(b"go\n")
        child.stdin.flush()
    except Exception:  # noqa: BLE001
        pass
    return _child_json(child, float(os.environ.get("PPRHIP_BENCH_CPU_S", "900")), "the CPU baseline child")


def cpu_share():
    """Cores this job may use at once: the cgroup's CPU quota where there is one (the GPU boxes give a one-GPU job 16
    of the host's 256 hardware threads; threads beyond the quota only get throttled), else the physical cores."""
    try:
        ids = set()
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    ids.add((phys, core))
                phys = core = None
        pcores = len(ids) or (os.cpu_count() or 1)
    except Exception:  # noqa: BLE001
        pcores = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            if quota:
                break
        except Exception:  # noqa: BLE001
            continue
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:  # noqa: BLE001
        aff = pcores
    share = int(max(1, min(pcores, aff, int(quota) if quota else pcores)))
    return share, pcores, quota


def cpu_baseline_child(args):
    """The reference's CPU path on this box's host cores; the Java itself cannot run here.
    Two measurements, one after the other:
      1. while the parent measures on the GPU (four threads of the job's share): the faithful port (hash maps, FIFO
         deque, hash set, the clock-driven loop with its 400 ns constant: oracle/ppr_baseline.cpp) on ONE thread, ONE
         query run to the end, every walk walked (Fora_Whole_Graph.java:93-140) - `value`; beside it, on three more
         threads, the dense-array port on three other sources, in full as well;
      2. after the parent's "go" (its GPU measurements are over): the array port on every core the job has, one query
         per core (walks thinned by --cpu-walk-divisor, times scaled back)."""
    try:
        from oracle import baseline as base
        from oracle import oracle as orc
        import threading
        pkg = importlib.import_module("personalized-pagerank-algorithms-on-neo4j_amd")
        host = load_host(pkg, args.scale)
        og = orc.OracleGraph(host.n, host.out_rp, host.out_ci, host.in_rp, host.in_ci)
        live = [int(s) for s in args.cpu_sources.split(",") if s]
        cores = base.hardware_threads()
        try:
            model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
        except Exception:  # noqa: BLE001
            model = "unknown"
        t_start = time.time()
        conf = og.conf_whole(ALPHA)
        _, omega = orc.fora_whole_params(conf, EPS)
        share, pcores, quota = cpu_share()
        # ---- 1. two faithful queries to the end (one thread each: the port's time varies by a factor of 1.7 with the
        # source and with what else runs on the shared host - 110 s in round 3, 193 s in round 4 - so one sample says
        # little) + the array port on two sources (two threads)
        box = {}

        def run_arrays():
            box["arr"] = base.fora_array_parallel(og, live[2:4], EPS, ALPHA, seed=3, walk_divisor=1, threads=2)

        def run_second():
            t1 = time.time()
            box["h2"] = base.fora_hashmap(og, live[1], EPS, ALPHA, seed=3, walk_divisor=1, push_budget_s=0.0)
            box["h2_wall"] = time.time() - t1

        th = [threading.Thread(target=run_arrays), threading.Thread(target=run_second)]
        for x in th:
            x.start()
        t0 = time.time()
        h = base.fora_hashmap(og, live[0], EPS, ALPHA, seed=3, walk_divisor=1, push_budget_s=0.0)
        h_wall = time.time() - t0
        for x in th:
            x.join()
        arr, h2 = box["arr"], box["h2"]
        h_query = h["push_s"] + h["walk_s"]
        h2_query = h2["push_s"] + h2["walk_s"]
        both = [h_query, h2_query]
        # ---- 2. every core of the job's share, one query each, once the parent's GPU measurements are over
        sys.stdin.readline()
        par_srcs = (live[4:4 + share] or live[:1])
        p = base.fora_array_parallel(og, par_srcs, EPS, ALPHA, seed=3, walk_divisor=args.cpu_walk_divisor, threads=share)
        # the run thinned the walks; per_query_s holds every query's time scaled to all of its walks, measured while the
        # other threads were running theirs: concurrent throughput = sum of the per-thread rates
        all_cores_qps = sum(1.0 / t for t in p["per_query_s"] if t > 0)
        pq = sorted(p["per_query_s"])
        wall = time.time() - t_start
        res = {
            "value": round(len(both) / sum(both), 6) if min(both) > 0 else None, "unit": "queries/s", "cores": 1,
            "kind": "port",
            "sample": "faithful (hash-map-shaped) port of Forward_Push/Fora_Whole_Graph/Monte_Carlo, TWO queries (sources %d "
                      "and %d) each run to the end on a thread of its own: %.1f s and %.1f s; value = queries per second "
                      "of one core = 2 / their sum.  The first: %d turn(s) of the clock-driven loop, %.1f s of pushes (%d "
                      "edge pushes, %.2f M/s), %d walks in %.1f s (%.2f M/s); beside them the dense-array port on 2 "
                      "sources in full (2 threads); then the array port on all %d cores of the job's share; %.0f s of "
                      "wall time in a background process, the first part beside the GPU measurements"
                      % (live[0], live[1], h_query, h2_query, h["rounds"], h["push_s"], h["edge_pushes"],
                         h["edge_pushes"] / h["push_s"] / 1e6 if h["push_s"] > 0 else 0.0, h["walks_run"], h["walk_s"],
                         h["walks_run"] / h["walk_s"] / 1e6 if h["walk_s"] > 0 else 0.0, share, wall),
            "seconds_per_query": round(sum(both) / len(both), 2),
            "seconds_per_query_each": [round(x, 2) for x in both], "sources": [int(live[0]), int(live[1])],
            "faithful_second": {"turns": h2["rounds"], "push_s": round(h2["push_s"], 2), "walk_s": round(h2["walk_s"], 2),
                                "edge_pushes": int(h2["edge_pushes"]), "walks": int(h2["walks_run"]),
                                "truncated": bool(h2["truncated"]), "wall_s": round(box["h2_wall"], 2)},
            "faithful": {"turns": h["rounds"], "push_s": round(h["push_s"], 2), "walk_s": round(h["walk_s"], 2),
                         "edge_pushes": int(h["edge_pushes"]), "walks": int(h["walks_run"]),
                         "walks_asked_for": int(h["walks_total"]), "truncated": bool(h["truncated"]),
                         "wall_s": round(h_wall, 2), "omega": omega,
                         "structures": "unordered_map<int64,double> x2, deque, unordered_set "
                                       "(HashMap/ConcurrentLinkedQueue/HashSet)"},
            "array": {"value": round(len(arr["per_query_s"]) / sum(arr["per_query_s"]), 5) if sum(arr["per_query_s"]) > 0 else None,
                      "cores": 1, "seconds_per_query_each": [round(x, 2) for x in arr["per_query_s"]],
                      "sources": [int(s) for s in live[2:4]],
                      "sample": "dense-array port, two sources in full (every walk), one thread each, while the two "
                                "faithful queries ran on two more"},
            "all_cores": {"value": round(all_cores_qps, 5), "cores": int(min(share, len(par_srcs))),
                          "queries": len(par_srcs), "wall_s_thinned_walks": round(p["wall_s"], 2),
                          "seconds_per_query_min_median_max": [round(pq[0], 1), round(pq[len(pq) // 2], 1), round(pq[-1], 1)],
                          "sample": "dense-array port, one query per core on %d live sources at once, on every core this "
                                    "job may use (cgroup quota %s of the host's %d physical cores / %d hardware threads; "
                                    "with 128 threads under the same quota the box delivered 0.13 queries/s, "
                                    "profiles/r03_bench_cpu128.json); every %d-th walk run, per-query times scaled to "
                                    "all walks" % (len(par_srcs), ("%.0f cores" % quota) if quota else "none", pcores, cores,
                                                   args.cpu_walk_divisor)},
            "host": {"nproc": cores, "physical_cores": pcores, "cpu_quota_cores": quota, "model": model},
        }
    except Exception as e:  # noqa: BLE001
        res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
