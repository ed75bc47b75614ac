// kernels_apbs.hip — All-Pair-Backward-Search: many targets in flight (gfx950).
//
// Base_Whole_Graph.preprocessing (Base_Whole_Graph.java:58-164) runs one backward search
// (Backward_Search.java:38-100) per target; the searches are independent.  A persistent workgroup takes one
// target at a time and keeps that target's whole state (residue, reserve, frontier) to itself:
//
//   tier 1  hash table in LDS (ds_cmpst / ds_add_rtn_f64).  Round 4 runs it in three steps on the device, because the tier
//           is bound by latency and barriers and not by anything a search computes (R-MAT 22: 257 ms for 3.6 M searches
//           on two workgroups per CU): a streaming pass that emits the targets without in-edges - 52 % of R-MAT 22's
//           nodes; their search is {t: 1.0} - and lists the others; a 512-slot table (20 KB: eight workgroups per
//           CU, ONE WAVE each: a search of a few hundred edges leaves four waves' worth of lanes idle, and every
//           wave of a workgroup executes every scan and barrier of the search) for the searches that touch fewer
//           than 384 nodes, another 20 % of the targets; the 2048-slot table
//           (76 KB: two per CU) for what the small one gave up.  Nine searches in ten touch fewer than 1536 nodes
//           (89 % of the targets, 8 % of the edges);
//   tier 2  the searches tier 1 gave up, on *dense* per-workgroup vectors (residue and reserve indexed by node id,
//           8n bytes each, zero between searches): an edge is ONE returning fp64 atomic and nothing else random;
//   tier 3  (host, fallback only) the engine's whole-vector backward search on the batch slots, for a search whose
//           frontier outgrows tier 2's lists or when the device cannot spare tier 2's vectors.
//
// Why dense vectors and not a bigger hash table (round 2 had a 524 288-slot table per workgroup in HBM): the edges
// live in the few searches that reach tens of thousands of nodes (tools/exp/apbs_census.c: 72 % of all edges in
// searches that touch more than 16 K nodes, and most nodes are touched once or twice), so their tables are far
// beyond L2 whatever their layout, and beyond L2 every random read-modify-write costs the same: 17-24 G per second
// chip-wide for an fp64 atomic, a compare-and-swap, a bypassing load followed by an atomic, or a plain load and
// store alike, for footprints from 25 MB to 8 GB (tools/micro/atomic_rate.hip, profiles/r03_atomic_rate.txt; fp
// atomics execute at the memory side and leave nothing in L2).  The hash table spent three to four such operations
// on an edge (key probe, claim, the source's out-degree, the add; 250 GB of line traffic per 2^18 targets); the
// dense vector spends one, the in-edge records carry the source's out-degree, and nothing is probed or claimed.
//
// Levels are frontier-synchronous exactly as in k_sparse_prepare / k_sparse_push: all frontier
// nodes give up their residue first, then their in-edges are expanded edge-parallel (degree
// prefix over sub-batches of frontier nodes), one atomic add per edge, crossing test on
// (old, old + add) with the reference's strict un-normalised threshold (Backward_Search.java:89).
// Entries with reserve >= threshold are appended as (source, target, pi) triples.
#include <algorithm>
#include <type_traits>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

constexpr int kApLdsCap = 2048;   // slots of the large LDS table (two workgroups per CU) ...
constexpr int kApFront = 512;
constexpr int kApSmallCap = 512;  // ... and of the small one (eight per CU)
constexpr int kApSmallFront = 128;
constexpr int kApSmallThreads = 64;

// in-edge record: the source of the edge and its out-degree (Backward_Search.java:84 divides by it per edge), so that
// an edge costs one coalesced 8-byte read instead of a column index and a gather of the source's row extent
struct InRec {
  int32_t u;
  uint32_t dout;
};

__global__ __launch_bounds__(256) void k_build_in_rec(const int32_t* __restrict__ in_ci,
                                                       const unsigned long long* __restrict__ out_ext,
                                                       unsigned long long m, InRec* __restrict__ rec) {
  for (unsigned long long e = blockIdx.x * 256ull + threadIdx.x; e < m; e += (unsigned long long)gridDim.x * 256ull) {
    const int32_t u = in_ci[e];
    rec[e] = InRec{u, (uint32_t)(out_ext[u] >> 32)};
  }
}

__device__ __forceinline__ InRec load_rec(const InRec* __restrict__ rec, size_t e) {
  // read once per search: non-temporal (the stream must not push anything useful out of the caches)
  const unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(rec) + e);
  return InRec{(int32_t)(uint32_t)w, (uint32_t)(w >> 32)};
}

// Workgroup exclusive scan of one value per thread for NW waves; scratch holds NW slots.
template <class T, int NW>
__device__ __forceinline__ T block_excl_scan_n(T x, T* scratch, T* total) {
  const int lane = lane_id(), wv = wave_id();
  T incl = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) scratch[wv] = incl;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const T sv = scratch[w];
    if (w < wv) base += sv;
    tot += sv;
  }
  __syncthreads();
  *total = tot;
  return base + incl - x;
}

// ------------------------------------------------------------------------------------------------
// tier 1: one target's state in an LDS hash table
// ------------------------------------------------------------------------------------------------
struct ApTable {  // one target's state in LDS
  int32_t* keys;   // node id or -1
  double* res;     // residue
  double* rsv;     // reserve
  double* pend;    // (1 - alpha) * residue taken at level start, per slot
  uint16_t* used;  // slots in insertion order
  uint16_t* cur;   // frontier (slots)
  uint16_t* nxt;
};

// returns the slot of node u, inserting it if absent; 0xFFFFFFFF when the table is full.  The insertion that takes
// the table past `limit` nodes raises *overflow at once: the search is handed to the next tier anyway, and the
// threads that see the flag stop inserting, so the table never fills up - in a full table every further lookup of a
// new node walks all of it (measured before this check: 4.3 M such walks of 2048 probes each, nine tenths of all
// the probes of the tier)
template <int CAP>
__device__ __forceinline__ uint32_t ap_slot(const ApTable& T, int32_t u, uint32_t* used_count, uint32_t limit,
                                            uint32_t* overflow) {
  constexpr uint32_t mask = CAP - 1;
  uint32_t s = ((uint32_t)u * 2654435761u) >> 7 & mask;
  for (uint32_t probes = 0; probes < (uint32_t)CAP; ++probes) {
    const int32_t k = T.keys[s];
    if (k == u) return s;
    if (k == -1) {
      const int32_t prev = atomicCAS(&T.keys[s], -1, u);  // ds_cmpst_rtn_b32
      if (prev == -1) {
        const uint32_t idx = atomicAdd(used_count, 1u);
        if (idx < (uint32_t)CAP) T.used[idx] = (uint16_t)s;
        if (idx >= limit) *overflow = 1;
        return s;
      }
      if (prev == u) return s;
    }
    s = (s + 1) & mask;
  }
  return 0xFFFFFFFFu;
}

// output side shared by both tiers: triple buffer, retry list, counters
struct ApOut {
  TripleRec* out_rec;
  unsigned long long out_cap;
  unsigned long long* out_count;
  unsigned long long* out_valid;
  int32_t* overflow_list;
  unsigned long long* overflow_count;
  unsigned long long* stat_pops;
  unsigned long long* stat_edges;
};

// target_list entries: a target id, or -(id + 1) for a target an earlier step has listed for another try (its record
// buffer was full); n_targets_dev != nullptr: the list's length is read there (written by the step before, on the device)
template <int CAP, int FRONT, int THREADS>
__global__ __launch_bounds__(THREADS) void k_apbs_lds(const int32_t* __restrict__ target_list, uint32_t t_begin,
                                                   uint32_t n_targets_arg, const unsigned long long* n_targets_dev,
                                                   unsigned long long* next_target,
                                                   const uint32_t* __restrict__ in_rp, const InRec* __restrict__ in_rec,
                                                   const int32_t* __restrict__ old2new, const int32_t* __restrict__ new2old,
                                                   double alpha, double rmax, ApOut O) {
  constexpr int kApLdsCap = CAP;  // (the names the body was written with)
  constexpr int kApFront = FRONT;
  static_assert(THREADS % 64 == 0 && (FRONT == THREADS || FRONT == 2 * THREADS),
                "the sub-batch code below stages one or two frontier entries per thread");
  __shared__ int32_t s_keys[kApLdsCap];
  __shared__ double s_res[kApLdsCap];
  __shared__ double s_rsv[kApLdsCap];
  __shared__ double s_pend[kApLdsCap];
  __shared__ uint16_t s_used[kApLdsCap];
  __shared__ uint16_t s_cur[kApLdsCap];
  __shared__ uint16_t s_nxt[kApLdsCap];
  __shared__ uint32_t f_row[kApFront];
  __shared__ uint32_t f_off[kApFront + 1];
  __shared__ double f_c[kApFront];
  const unsigned long long n_targets = n_targets_dev ? *n_targets_dev : (unsigned long long)n_targets_arg;
  __shared__ uint32_t s_scan[THREADS / 64];
  __shared__ unsigned long long s_scan64[THREADS / 64];
  __shared__ uint32_t s_used_count, s_nnext, s_overflow;
  __shared__ unsigned long long s_t, s_out_base, s_tot;
  const int tid = threadIdx.x;

  ApTable T{s_keys, s_res, s_rsv, s_pend, s_used, s_cur, s_nxt};
  const uint32_t limit = kApLdsCap - kApLdsCap / 4;  // give up at 75 % load
  for (uint32_t i = tid; i < (uint32_t)kApLdsCap; i += THREADS) {
    T.keys[i] = -1;
    T.res[i] = 0.0;
    T.rsv[i] = 0.0;
  }
  __syncthreads();
  unsigned long long pops = 0, edges = 0;

  for (;;) {
    if (tid == 0) s_t = atomic_add_u64(next_target, 1ull);
    __syncthreads();
    const unsigned long long ti = s_t;
    if (ti >= n_targets) break;
    int32_t t_old = target_list ? target_list[ti] : (int32_t)(t_begin + ti);
    if (t_old < 0) t_old = -(t_old + 1);
    const int32_t t = old2new[t_old];
    if (tid == 0) {
      s_used_count = 0;
      s_overflow = 0;
      // the record buffer is full (somebody's entries did not fit): a search started now could not hand over its
      // entries either, so it is not run but listed for the host's next pass
      s_tot = (__hip_atomic_load(O.out_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ~0ull) ? 1ull : 0ull;
      if (s_tot) O.overflow_list[atomic_add_u64(O.overflow_count, 1ull)] = -(t_old + 1);
    }
    __syncthreads();
    if (s_tot) {
      __syncthreads();
      continue;
    }
    uint32_t nf = 0;
    if (in_rp[t + 1] == in_rp[t]) {  // Backward_Search.java:46-49: reserve = {t: 1.0}
      if (tid == 0) {
        const uint32_t s = ap_slot<CAP>(T, t, &s_used_count, limit, &s_overflow);
        T.rsv[s] = 1.0;
      }
    } else {
      if (tid == 0) {
        const uint32_t s = ap_slot<CAP>(T, t, &s_used_count, limit, &s_overflow);
        T.res[s] = 1.0;  // :54-56; the target is pushed unconditionally first
        T.cur[0] = (uint16_t)s;
      }
      nf = 1;
    }
    __syncthreads();

    while (nf > 0 && !s_overflow) {
      // ---- every frontier node gives up its residue (:58-67,72)
      for (uint32_t i = tid; i < nf; i += THREADS) {
        const uint32_t s = T.cur[i];
        const double rc = T.res[s];
        T.res[s] = 0.0;
        T.rsv[s] = T.rsv[s] + rc * alpha;
        T.pend[s] = (1.0 - alpha) * rc;
      }
      if (tid == 0) s_nnext = 0;
      pops += (tid == 0) ? nf : 0;
      __syncthreads();
      // ---- in-edges of the frontier, 512 frontier nodes at a time
      for (uint32_t fb = 0; fb < nf && !s_overflow; fb += kApFront) {
        const uint32_t cnt = nf - fb < (uint32_t)kApFront ? nf - fb : (uint32_t)kApFront;
        uint32_t d0 = 0, d1 = 0;
        {
          const uint32_t i0 = tid, i1 = tid + THREADS;
          if (i0 < cnt) {
            const uint32_t s = T.cur[fb + i0];
            const int32_t v = T.keys[s];
            const uint32_t b = in_rp[v];
            d0 = in_rp[v + 1] - b;
            f_row[i0] = b;
            f_c[i0] = T.pend[s];
          }
          if (i1 < cnt) {
            const uint32_t s = T.cur[fb + i1];
            const int32_t v = T.keys[s];
            const uint32_t b = in_rp[v];
            d1 = in_rp[v + 1] - b;
            f_row[i1] = b;
            f_c[i1] = T.pend[s];
          }
        }
        // exclusive prefix of the degrees over the sub-batch (two elements per thread: i, i + THREADS)
        uint32_t tot0 = 0, tot1 = 0, e1 = 0;
        const uint32_t e0 = block_excl_scan_n<uint32_t, THREADS / 64>(d0, s_scan, &tot0);
        if (FRONT > THREADS) e1 = block_excl_scan_n<uint32_t, THREADS / 64>(d1, s_scan, &tot1);  // (the small table stages 128 entries)
        if ((uint32_t)tid < cnt) f_off[tid] = e0;
        if ((uint32_t)tid + THREADS < cnt) f_off[tid + THREADS] = tot0 + e1;
        if (tid == 0) f_off[cnt] = tot0 + tot1;
        __syncthreads();
        const uint32_t E = f_off[cnt];
        edges += (tid == 0) ? E : 0;
        // four edges per thread in flight: edge records first, then table updates
        for (uint32_t base = tid; base < E; base += 4 * THREADS) {
          if (s_overflow) break;  // the search goes to the next step anyway: do not read the rest of a hub's in-edges
          InRec rc4[4];
          double cc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t e = base + (uint32_t)THREADS * q;
            rc4[q] = InRec{-1, 1u};
            cc[q] = 0.0;
            if (e < E) {
              uint32_t lo = 0, hi = cnt;  // last frontier entry whose edge range starts at or before e
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (f_off[mid] <= e) lo = mid + 1; else hi = mid;
              }
              const uint32_t i = lo - 1;
              rc4[q] = load_rec(in_rec, (size_t)f_row[i] + (e - f_off[i]));
              cc[q] = f_c[i];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (rc4[q].u < 0 || s_overflow) continue;
            const double add = cc[q] / (double)rc4[q].dout;  // :84-85
            const uint32_t s = ap_slot<CAP>(T, rc4[q].u, &s_used_count, limit, &s_overflow);
            if (s == 0xFFFFFFFFu) {
              s_overflow = 1;
              continue;
            }
            const double old = atomic_add_ret(&T.res[s], add);  // ds_add_rtn_f64
            const double nw = old + add;
            if (!(old > rmax) && nw > rmax) {  // :89 strict, un-normalised; first crossing of the level
              const uint32_t pos = atomicAdd(&s_nnext, 1u);
              if (pos < (uint32_t)kApLdsCap) T.nxt[pos] = (uint16_t)s;
            }
          }
        }
        __syncthreads();
        if (tid == 0 && s_used_count > limit) s_overflow = 1;
        __syncthreads();
      }
      nf = s_nnext;
      uint16_t* tmp = T.cur; T.cur = T.nxt; T.nxt = tmp;
      __syncthreads();
    }

    const uint32_t used = s_used_count < (uint32_t)kApLdsCap ? s_used_count : (uint32_t)kApLdsCap;
    const bool ovf = s_overflow != 0;
    // ---- emit entries >= threshold (Base_Whole_Graph.java:80-88)
    bool retry = ovf;
    if (!ovf) {
      unsigned long long run = 0;
      for (uint32_t c0 = 0; c0 < used; c0 += THREADS) {  // count first, one reservation per target
        const uint32_t i = c0 + tid;
        const bool take = i < used && T.rsv[T.used[i]] > 0.0 && T.rsv[T.used[i]] >= rmax;
        run += take ? 1ull : 0ull;
      }
      const unsigned long long total = block_sum_u64(run, s_scan64);  // valid in thread 0
      if (tid == 0) {
        s_tot = total;
        s_out_base = total ? atomic_add_u64(O.out_count, total) : 0ull;
      }
      __syncthreads();
      const unsigned long long tot = s_tot;
      if (s_out_base + tot > O.out_cap) {
        retry = true;  // the triple buffer is full: the host drains it and runs this target again
        if (tid == 0) atomicMin(O.out_valid, s_out_base);
      } else {
        unsigned long long at = s_out_base;
        for (uint32_t c0 = 0; c0 < used; c0 += THREADS) {
          const uint32_t i = c0 + tid;
          const uint32_t s = i < used ? T.used[i] : 0u;
          const bool take = i < used && T.rsv[s] > 0.0 && T.rsv[s] >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex =
              block_excl_scan_n<unsigned long long, THREADS / 64>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) {
            O.out_rec[at + ex] = TripleRec{new2old[T.keys[s]], t_old, T.rsv[s]};
          }
          at += chunk_total;
        }
      }
    }
    if (retry && tid == 0) {  // table overflow: +t, triple buffer full: -(t + 1)
      const unsigned long long p = atomic_add_u64(O.overflow_count, 1ull);
      O.overflow_list[p] = ovf ? t_old : -(t_old + 1);
    }
    __syncthreads();
    // ---- clear the touched slots (all of them after an overflow)
    if (ovf) {
      for (uint32_t i = tid; i < (uint32_t)kApLdsCap; i += THREADS) {
        T.keys[i] = -1;
        T.res[i] = 0.0;
        T.rsv[i] = 0.0;
      }
    } else {
      for (uint32_t i = tid; i < used; i += THREADS) {
        const uint32_t s = T.used[i];
        T.keys[s] = -1;
        T.res[s] = 0.0;
        T.rsv[s] = 0.0;
      }
    }
    __syncthreads();
  }
  const unsigned long long ps = block_sum_u64(pops, s_scan64);
  const unsigned long long es = block_sum_u64(edges, s_scan64);
  if (tid == 0) {
    if (ps) atomic_add_u64(O.stat_pops, ps);
    if (es) atomic_add_u64(O.stat_edges, es);
  }
}

// Tier 1, first step: the targets [t_begin, t_begin + n) in one streaming pass.  A target without in-edges yields
// {t: 1.0} (Backward_Search.java:46-49: emitted here when 1.0 >= threshold, one reservation per wave).  The others
// are routed by their in-degree, which predicts the size of a search well (tools/exp: of R-MAT 22's targets with 1-2
// in-edges 64-80 % touch fewer than 384 nodes; with 5-8, 74 % fewer than 1536; with 17 and more, 94-100 % outgrow the
// LDS tables): below deg_big to the small table's list, below deg_dense to the large table's, the rest straight to the
// give-up list the host hands to the dense tier - a search that starts in a table it is going to outgrow is work done
// twice.  A wrong guess is only that: every table still hands on what it cannot hold.  A record that finds the buffer
// full sends its target to the large table's list as -(t + 1) (the table kernel's first branch is the same rule).
__global__ __launch_bounds__(256) void k_apbs_split(uint32_t t_begin, uint32_t n, const uint32_t* __restrict__ in_rp,
                                                     const int32_t* __restrict__ old2new, double rmax, ApOut O,
                                                     int32_t* __restrict__ list_small, unsigned long long* __restrict__ n_small,
                                                     int32_t* __restrict__ list_big, unsigned long long* __restrict__ n_big,
                                                     uint32_t deg_big, uint32_t deg_dense) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t i0 = blockIdx.x * 256u; i0 < n; i0 += stride) {  // (wave-uniform trip count)
    const uint32_t i = i0 + threadIdx.x;
    const int32_t t_old = (int32_t)(t_begin + i);
    uint32_t deg = 0;
    if (i < n) {
      const int32_t t = old2new[t_old];
      deg = in_rp[t + 1] - in_rp[t];
    }
    const bool trivial = i < n && deg == 0;
    int32_t entry = t_old;
    int route = (i >= n || trivial) ? -1 : (deg < deg_big ? 0 : (deg < deg_dense ? 1 : 2));
    const bool emit = trivial && 1.0 >= rmax;
    const unsigned long long em = __ballot(emit);
    if (em) {
      unsigned long long base = 0;
      if (lane == 0) base = atomic_add_u64(O.out_count, (unsigned long long)__popcll(em));
      base = __shfl(base, 0);
      const unsigned long long pos = base + (unsigned long long)__popcll(em & ((1ull << lane) - 1ull));
      if (emit) {
        if (pos < O.out_cap) {
          O.out_rec[pos] = TripleRec{t_old, t_old, 1.0};
        } else {  // buffer full: the host drains it and the target runs again
          atomicMin(O.out_valid, base < O.out_cap ? O.out_cap : base);
          route = 1;
          entry = -(t_old + 1);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const unsigned long long lm = __ballot(route == r);
      if (!lm) continue;
      unsigned long long* counter = r == 0 ? n_small : (r == 1 ? n_big : O.overflow_count);
      int32_t* list = r == 0 ? list_small : (r == 1 ? list_big : O.overflow_list);
      unsigned long long base = 0;
      if (lane == 0) base = atomic_add_u64(counter, (unsigned long long)__popcll(lm));
      base = __shfl(base, 0);
      if (route == r) list[base + (unsigned long long)__popcll(lm & ((1ull << lane) - 1ull))] = entry;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tier 2: one target's state in dense per-workgroup vectors, big levels shared between workgroups
// ------------------------------------------------------------------------------------------------
// Workspace of one workgroup (HBM, vectors all-zero between searches): residue[n], reserve[n], and the lists a search
// keeps: the nodes whose residue it has made non-zero (for the clean-up), the current / next frontier, per frontier
// position the pending contribution, the first in-edge and the exclusive prefix of the in-degrees (the level's edge
// space), per chunk of that edge space the frontier position it starts in, and the nodes popped at least once (the
// only ones that can hold a reserve).
//
// Work sharing.  The edges are unevenly spread: a handful of searches push millions of edges (one of 2^18 targets of
// an R-MAT 22 pushes 25 M) and one workgroup moves ~0.3 G edges/s however idle the rest of the chip is, so such a
// search alone set the kernel's duration (50 of 82 ms with every other workgroup done).  A level's edge space is
// therefore cut into chunks of kDnChunk edges; a level of at least kDnShareMin chunks is *posted* on the owner's
// board entry, and workgroups that have run out of targets take chunks of posted levels (compare-and-swap on
// {sequence, next chunk}) - the residue updates are agent-scope atomics on the owner's vector, the appends go through
// counters on the owner's board entry, and everything a helper reads or writes in the owner's lists bypasses the
// (per-XCD, mutually incoherent) L2s.  The owner takes chunks of its own level like everybody else and goes on when
// all of them are done; a helper never waits for anything, so nobody can wait in a cycle.
constexpr int kDnThreads = 1024;
constexpr int kDnWaves = kDnThreads / 64;
constexpr int kDnStage = 1024;      // frontier entries staged in LDS at a time
constexpr int kDnIlp = 4;           // edges a thread keeps in flight
constexpr uint32_t kDnChunkDefault = 32768;  // edges per chunk of a level's edge space (tests shrink it)
constexpr uint32_t kDnShareMin = 4;          // levels of at least this many chunks are posted
constexpr int kDnHotDefault = 8192;          // ids whose residue and reserve live in the owner's LDS (16 bytes each)
constexpr int kDnHotMax = 8192;
constexpr unsigned long long kDnWaitTicks = 3000000000ull;  // 30 s of the 100 MHz clock: no wait in the kernel is longer
uint32_t apbs_default_chunk() { return kDnChunkDefault; }

struct DnBoard {  // one per workgroup, 64 bytes
  unsigned long long next;  // sequence << 32 | next chunk; the level is open to helpers while the sequence is odd
  uint32_t n_chunks, nf, which, done;  // which: frontier list the level reads (0 / 1)
  uint32_t tcount, nnext, giveup, pad0;
  unsigned long long pad1[3];
};
static_assert(sizeof(DnBoard) == 64, "one board entry per 64-byte block");

struct DenseWs {
  double* res;
  double* rsv;
  double* pend;      // [cap_f] pending contribution, by frontier position
  uint32_t* frow;    // [cap_f] first in-edge of the frontier node
  uint32_t* eoff;    // [cap_f + 1] exclusive prefix of the frontier's in-degrees
  uint32_t* cstart;  // [cap_c] frontier position in which chunk c of the edge space starts
  int32_t* touched;  // [cap_t]
  int32_t* fr[2];    // [cap_f] frontier lists
  int32_t* plist;    // [cap_f] nodes popped for the first time, in pop order
};

// sizes of one workspace: n nodes, list capacities, cap_c = chunks a level can have (m / chunk + 2)
struct DnDims {
  uint32_t n, cap_t, cap_f, cap_c, chunk;
};
__host__ __device__ inline size_t dense_ws_bytes(const DnDims& D) {
  return ((size_t)8 * (2 * (size_t)D.n + D.cap_f) +
          (size_t)4 * ((size_t)D.cap_t + 5 * (size_t)D.cap_f + 1 + D.cap_c) + 255) & ~(size_t)255;
}
static DnDims dims_of(uint32_t n, unsigned long long m, uint32_t cap_t, uint32_t cap_f, uint32_t chunk) {
  return DnDims{n, cap_t, cap_f, (uint32_t)(m / chunk) + 2u, chunk};
}
size_t apbs_dense_bytes(uint32_t n, unsigned long long m, uint32_t cap_t, uint32_t cap_f, uint32_t chunk) {
  return dense_ws_bytes(dims_of(n, m, cap_t, cap_f, chunk));
}
size_t apbs_board_bytes(uint32_t blocks) { return sizeof(DnBoard) * (size_t)blocks + 64; }

__device__ __forceinline__ DenseWs dense_ws_of(char* ws_base, uint32_t wg, const DnDims& D) {
  DenseWs W;
  char* base = ws_base + (size_t)wg * dense_ws_bytes(D);
  W.res = (double*)base;
  W.rsv = W.res + D.n;
  W.pend = W.rsv + D.n;
  W.frow = (uint32_t*)(W.pend + D.cap_f);
  W.eoff = W.frow + D.cap_f;
  W.cstart = W.eoff + D.cap_f + 1;
  W.touched = (int32_t*)(W.cstart + D.cap_c);
  W.fr[0] = W.touched + D.cap_t;
  W.fr[1] = W.fr[0] + D.cap_f;
  W.plist = W.fr[1] + D.cap_f;
  return W;
}

// Every access to the vectors and lists bypasses L1 / L2 (agent-scope relaxed = sc1): the residue vector is updated by
// fp64 atomics, which execute at the memory side and keep nothing in L2, and the lists of a posted level are read and
// extended by workgroups on other XCDs, whose L2s are not coherent with the owner's.
template <class T>
__device__ __forceinline__ T dn_load(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void dn_store(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A word that is also the target of atomic read-modify-writes (the board's counters) is (re)set by an atomic exchange
// whose result is waited for: a store followed by an atomic on the same address is not guaranteed to arrive at the
// memory side in that order, and an atomic that overtakes the reset would act on the value from before it.
template <class T>
__device__ __forceinline__ void dn_reset(T* p, T v) {
  (void)__hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // the exchange (and every store before it) has been acknowledged before anything that follows is issued.  No cache
  // maintenance is needed anywhere in this protocol: everything workgroups hand to each other is written and read
  // with accesses that bypass L1 / L2 (sc1), so only the ORDER of those accesses matters.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Workgroup barrier that also waits for the global stores and atomics of every wave: __syncthreads() alone waits for
// LDS only (s_waitcnt lgkmcnt(0); the compiler's memory model lets global accesses of one CU order themselves), which
// is not enough where lane 0 goes on to tell ANOTHER workgroup that the lists the other waves have just written are
// ready - every storing wave has to have its own stores acknowledged first (they are write-through: sc1).
__device__ __forceinline__ void dn_barrier_global() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

struct DnLds {  // staging of one workgroup (owner or helper): up to kDnStage frontier entries
  uint32_t eoff[kDnStage + 1];  // exclusive prefix of the in-degrees (absolute positions in the level's edge space)
  uint32_t row[kDnStage];       // first in-edge
  double c[kDnStage];           // pending contribution
  uint32_t i0;  // frontier position a chunk starts in (read by one lane, so that every wave sees the same value)
  uint32_t wtot[kDnWaves], wbase;  // shared levels: the waves' append counts of one pass, and the block reserved for them
};

// where a search's appends are counted: in LDS while the search stays on its own workgroup, on the owner's board entry
// (global atomics) while a level is shared
struct DnSinks {
  uint32_t* tcount;  // nodes listed for the clean-up
  uint32_t* nnext;   // next frontier
  uint32_t* giveup;  // a list is full
};

// Ids below `hot_n` - in the internal order the nodes with the most out-edges, the destinations of a third of all
// in-edges (tools/exp/apbs_levels.c: 34 % of a dense-tier search's edges end below 8 K, 42 % below 16 K) - have their
// residue and reserve in the OWNER's LDS for the whole search (hres / hrsv): an edge to one of them is a returning
// ds_add_f64 instead of a memory-side read-modify-write, it needs no entry in the clean-up list, and a pop of one
// reads LDS.  Only the owner can reach its LDS, so a level that is posted for helpers moves the hot residues into
// the (all-zero) head of the global vector first and back afterwards (dn_hot_spill / dn_hot_fill); while a level is
// shared, hot and cold destinations alike take the memory-side atomic.
//
// Edges [lo, hi) of a level's edge space, which the `cnt` entries staged in L cover: every edge is ONE returning atomic
// on the residue vector; residues that leave zero and nodes that cross the threshold are appended to the search's
// lists.  Called by all threads of a workgroup with workgroup-uniform arguments.
// kWgAgg (shared levels, whose counters are global words that every helper of the owner adds to): the workgroup's
// waves reserve their clean-up slots with ONE atomic per pass instead of one each; hres == nullptr there.
template <bool kWgAgg>
__device__ __forceinline__ void dn_edge_range(const DenseWs& W, DnLds& L, uint32_t cnt, uint32_t lo, uint32_t hi,
                                              int32_t* nxt, const DnSinks& S, const InRec* __restrict__ in_rec,
                                              double rmax, uint32_t cap_t, uint32_t cap_f, double* hres, uint32_t hot_n) {
  const int tid = threadIdx.x, lane = lane_id();
  // wave-uniform trip count (the appends below are wave-aggregated)
  for (unsigned long long base0 = lo; base0 < hi; base0 += kDnThreads * kDnIlp) {
    InRec rc4[kDnIlp];
    double add[kDnIlp], old[kDnIlp];
#pragma unroll
    for (int q = 0; q < kDnIlp; ++q) {
      const unsigned long long e64 = base0 + (unsigned long long)(kDnThreads * q + tid);
      const uint32_t e = (uint32_t)e64;
      rc4[q] = InRec{-1, 1u};
      add[q] = 0.0;
      if (e64 < hi) {
        uint32_t a = 0, z = cnt;  // last staged entry whose edge range starts at or before e
        while (a < z) {
          const uint32_t mid = (a + z) >> 1;
          if (L.eoff[mid] <= e) a = mid + 1; else z = mid;
        }
        const uint32_t i = a - 1;
        rc4[q] = load_rec(in_rec, (size_t)L.row[i] + (e - L.eoff[i]));
        add[q] = L.c[i];
      }
    }
#pragma unroll
    for (int q = 0; q < kDnIlp; ++q) {
      add[q] = add[q] / (double)rc4[q].dout;  // :84-85, every edge's quotient rounds on its own
      old[q] = 0.0;
      if (rc4[q].u >= 0) {
        if (!kWgAgg && (uint32_t)rc4[q].u < hot_n) old[q] = atomic_add_ret(&hres[rc4[q].u], add[q]);  // ds_add_rtn_f64
        else old[q] = atomic_add_ret(&W.res[rc4[q].u], add[q]);
      }
    }
    // a residue that leaves zero: the node is remembered for the clean-up (a node popped in between is listed twice,
    // which only clears it twice); hot ids are cleared as a range.  One reservation per wave for the kDnIlp edges of
    // its lanes.
    unsigned long long fm[kDnIlp];
    uint32_t tot = 0;
#pragma unroll
    for (int q = 0; q < kDnIlp; ++q) {
      fm[q] = __ballot(rc4[q].u >= 0 && (uint32_t)rc4[q].u >= hot_n && old[q] == 0.0);
      tot += (uint32_t)__popcll(fm[q]);
    }
    uint32_t wb = 0;
    if (kWgAgg) {
      if (lane == 0) L.wtot[wave_id()] = tot;
      __syncthreads();
      uint32_t all = 0;
#pragma unroll
      for (int w = 0; w < kDnWaves; ++w) {
        const uint32_t x = L.wtot[w];
        wb += w < wave_id() ? x : 0u;
        all += x;
      }
      if (tid == 0) L.wbase = all ? atomicAdd(S.tcount, all) : 0u;
      __syncthreads();
      wb += L.wbase;
    } else if (tot) {
      if (lane == 0) wb = atomicAdd(S.tcount, tot);
      wb = __shfl(wb, 0);
    }
    if (tot) {
#pragma unroll
      for (int q = 0; q < kDnIlp; ++q) {
        if ((fm[q] >> lane) & 1ull) {
          const uint32_t pos = wb + (uint32_t)__popcll(fm[q] & ((1ull << lane) - 1ull));
          if (pos < cap_t) dn_store(&W.touched[pos], rc4[q].u);
        }
        wb += (uint32_t)__popcll(fm[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < kDnIlp; ++q)
      if (rc4[q].u >= 0 && !(old[q] > rmax) && old[q] + add[q] > rmax) {  // :89 strict, un-normalised; first crossing
        const uint32_t pos = atomicAdd(S.nnext, 1u);
        if (pos < cap_f) dn_store(&nxt[pos], rc4[q].u);
        else atomicOr(S.giveup, 1u);
      }
  }
}

// The owner's hot residues move into the head of its global vector before a level is posted for helpers (the head is
// all-zero: hot ids never take the global atomic while a search is on its own), and back - leaving the head zero again -
// when the level has closed.  Called by all threads of the owner's workgroup.
__device__ __forceinline__ void dn_hot_spill(const DenseWs& W, double* hres, uint32_t hot_n) {
  for (uint32_t s = threadIdx.x; s < hot_n; s += kDnThreads) {
    const double v = hres[s];
    if (v != 0.0) {
      dn_store(&W.res[s], v);
      hres[s] = 0.0;
    }
  }
}
__device__ __forceinline__ void dn_hot_fill(const DenseWs& W, double* hres, uint32_t hot_n) {
  for (uint32_t s = threadIdx.x; s < hot_n; s += kDnThreads) {
    const double v = dn_load(&W.res[s]);
    if (v != 0.0) {
      hres[s] = v;
      dn_store(&W.res[s], 0.0);
    }
  }
}

// One chunk [c * chunk, ...) of a POSTED level's edge space, read from the owner's level layout in HBM; appends through
// the owner's board entry.  Called by all threads of a workgroup (the owner's or a helper's); which / nf / E / c must
// be workgroup-uniform (the callers hand them round through LDS): the loops below hold barriers.
__device__ __forceinline__ void dn_chunk(const DenseWs& W, DnBoard* B, uint32_t which, uint32_t nf, uint32_t E,
                                         uint32_t c, const InRec* __restrict__ in_rec, double rmax, const DnDims& D,
                                         DnLds& L, uint32_t hot_n) {
  const int tid = threadIdx.x;
  const uint32_t chunk_lo = c * D.chunk;
  const uint32_t chunk_hi = (E - chunk_lo > D.chunk) ? chunk_lo + D.chunk : E;
  if (tid == 0) L.i0 = dn_load(&W.cstart[c]);
  __syncthreads();
  uint32_t i0 = L.i0;
  uint32_t ce = chunk_lo;
  const DnSinks S{&B->tcount, &B->nnext, &B->giveup};
  while (ce < chunk_hi) {
    const uint32_t cnt = nf - i0 < (uint32_t)kDnStage ? nf - i0 : (uint32_t)kDnStage;
    if (cnt == 0 || i0 >= nf) break;  // (cannot happen: eoff[nf] = E > ce)
    __syncthreads();                   // the staging arrays of the round before are no longer read
    if ((uint32_t)tid < cnt) {
      L.eoff[tid] = dn_load(&W.eoff[i0 + tid]);
      L.row[tid] = dn_load(&W.frow[i0 + tid]);
      L.c[tid] = dn_load(&W.pend[i0 + tid]);
    }
    if (tid == 0) L.eoff[cnt] = dn_load(&W.eoff[i0 + cnt]);
    __syncthreads();
    const uint32_t cov_hi = L.eoff[cnt] < chunk_hi ? L.eoff[cnt] : chunk_hi;
    if (cov_hi <= ce) break;  // (cannot happen: the staged entries cover ce; a guard against looping on bad data)
    dn_edge_range<true>(W, L, cnt, ce, cov_hi, W.fr[which ^ 1], S, in_rec, rmax, D.cap_t, D.cap_f, nullptr, hot_n);
    ce = cov_hi;
    i0 += cnt;
  }
  dn_barrier_global();  // every wave's appends have arrived before lane 0 reports the chunk done
}

struct DnHelpLds {  // a workgroup's look at the boards (dn_help_once)
  uint32_t owner, chunk, edges, nf, which, open, done;
};

// One look at the boards by a workgroup that holds no search: the nearest open level after its own entry, one chunk of
// it pushed into the owner's vectors.  Returns 2 when a chunk was pushed, 1 when somebody else was faster (look again),
// 0 when nothing is open; H.done says whether every target of the launch is finished (or the launch is aborted).
// Every path ends with a barrier behind its last lane-0 block (see the note at the owner's chunk loop).
__device__ __forceinline__ int dn_help_once(DnHelpLds& H, DnLds& L, DnBoard* board, char* ws_base, const DnDims& D,
                                            const InRec* __restrict__ in_rec, double rmax, uint32_t hot_n,
                                            uint32_t n_owners, const unsigned long long* n_open,
                                            const unsigned long long* done_targets, uint32_t n_targets,
                                            const unsigned long long* abort_word) {
  const int tid = threadIdx.x;
  if (tid == 0) {
    H.owner = 0xFFFFFFFFu;  // owner to help
    H.done = (dn_load(done_targets) >= (unsigned long long)n_targets || dn_load(abort_word)) ? 1u : 0u;
    H.open = (dn_load(n_open) != 0ull && !dn_load(abort_word)) ? 1u : 0u;
  }
  __syncthreads();
  // (one word polled while nothing is posted: the board is only scanned when somebody has a level open)
  if (!H.open) {
    __syncthreads();  // (H is not rewritten before every wave has read it)
    return 0;
  }
  if ((uint32_t)tid < n_owners && (uint32_t)tid != blockIdx.x) {
    const unsigned long long nx = dn_load(&board[tid].next);
    // the nearest open entry after this workgroup's own: helpers spread over the owners instead of all taking the
    // first one (every append of a shared level is an atomic on the owner's counters)
    if (((nx >> 32) & 1ull) && (uint32_t)nx < dn_load(&board[tid].n_chunks))
      atomicMin(&H.owner, ((uint32_t)tid + gridDim.x - blockIdx.x) % gridDim.x);
  }
  __syncthreads();
  const uint32_t owner = H.owner == 0xFFFFFFFFu ? 0xFFFFFFFFu : (H.owner + blockIdx.x) % gridDim.x;
  __syncthreads();
  if (owner == 0xFFFFFFFFu) return 0;
  DnBoard* OB = board + owner;
  if (tid == 0) {
    // the level's parameters are valid for the sequence read with them iff the compare-and-swap on
    // {sequence, chunk} succeeds: the owner rewrites them only while the entry is closed
    const unsigned long long nx = dn_load(&OB->next);
    const uint32_t nch = dn_load(&OB->n_chunks);
    H.nf = dn_load(&OB->nf);
    H.which = dn_load(&OB->which);
    H.edges = dn_load(&dense_ws_of(ws_base, owner, D).eoff[H.nf <= D.cap_f ? H.nf : 0u]);
    uint32_t c = 0xFFFFFFFFu;
    if (((nx >> 32) & 1ull) && (uint32_t)nx < nch) {
      unsigned long long expect = nx;
      if (__hip_atomic_compare_exchange_strong(&OB->next, &expect, nx + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT)) {
        c = (uint32_t)nx;
      }
    }
    H.chunk = c;
  }
  __syncthreads();
  const uint32_t c = H.chunk, onf = H.nf, owhich = H.which, oE = H.edges;
  __syncthreads();
  if (c == 0xFFFFFFFFu) return 1;  // somebody else was faster: look again
  const DenseWs OW = dense_ws_of(ws_base, owner, D);
  dn_chunk(OW, OB, owhich & 1u, onf, oE, c, in_rec, rmax, D, L, hot_n);
  if (tid == 0) atomicAdd(&OB->done, 1u);  // (dn_chunk ended with every wave's stores acknowledged)
  __syncthreads();                         // (lane-0 blocks never sit next to a back edge)
  return 2;
}

__global__ __launch_bounds__(kDnThreads) void k_apbs_dense(const int32_t* __restrict__ target_list, uint32_t n_targets,
                                                            unsigned long long* next_target,
                                                            const uint32_t* __restrict__ in_rp,
                                                            const InRec* __restrict__ in_rec,
                                                            const int32_t* __restrict__ old2new,
                                                            const int32_t* __restrict__ new2old, double alpha,
                                                            double rmax, ApOut O, char* ws_base, DnBoard* board,
                                                            unsigned long long* done_targets,
                                                            unsigned long long* n_open,
                                                            unsigned long long* abort_word, DnDims D, int share,
                                                            uint32_t n_owners, uint32_t hot_n, int help_between,
                                                            unsigned long long* __restrict__ dbg) {
  // n_owners: the workgroups 0 .. n_owners - 1 have a workspace and take targets; the others only help with posted
  // levels (the pass for the few searches whose lists need room for every node: a handful of full-size workspaces,
  // the whole chip working on their levels)
  // dbg (developer switch PPRHIP_APBS_DEBUG, else nullptr; HOST memory): per workgroup {searches, edges of its own
  // searches, ticks of the 100 MHz clock spent in pops + scans, own edges, waiting for helpers, emission, clean-up,
  // helping / idle, the tick at which the workgroup ended, and where it is: stage << 32 | detail (printed by the
  // host's watchdog when a launch does not come back)}
  // hot_n: ids below it keep residue and reserve in this workgroup's LDS while it owns a search (dn_edge_range);
  // dynamic LDS: hres[hot_n], hrsv[hot_n]
  extern __shared__ double dn_hot[];
  double* const hres = dn_hot;
  double* const hrsv = dn_hot + hot_n;
  __shared__ DnLds L;
  __shared__ uint32_t s_scan[kDnWaves];
  __shared__ unsigned long long s_scan64[kDnWaves];
  __shared__ uint32_t s_pcount, s_carry, s_job[4], s_gaveup;
  __shared__ DnHelpLds H;
  __shared__ uint32_t s_tcount, s_nnext, s_giveup;  // the search's append counters while it stays on this workgroup
  __shared__ unsigned long long s_t, s_out_base, s_tot;
  const int tid = threadIdx.x;
  const uint32_t n = D.n, cap_t = D.cap_t, cap_f = D.cap_f, chunk = D.chunk;
  const DenseWs W = dense_ws_of(ws_base, blockIdx.x, D);
  DnBoard* B = board + blockIdx.x;
  const DnSinks S_local{&s_tcount, &s_nnext, &s_giveup};
  unsigned long long pops = 0, edges = 0;
  unsigned long long tk[6] = {0, 0, 0, 0, 0, 0}, t_mark = dbg ? wall_clock64() : 0ull, n_search = 0;
#define DN_TICK(i)                                   \
  if (dbg && tid == 0) {                             \
    const unsigned long long now_ = wall_clock64();  \
    tk[i] += now_ - t_mark;                          \
    t_mark = now_;                                   \
  }
#define DN_AT(stage, detail)                                                                                  \
  if (dbg && tid == 0)                                                                                        \
    __hip_atomic_store(dbg + (size_t)blockIdx.x * 12 + 9, ((unsigned long long)(stage) << 32) | (uint32_t)(detail), \
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  uint32_t seq = 0;  // even: this workgroup's board entry is closed
  for (uint32_t i = tid; i < 2 * hot_n; i += kDnThreads) dn_hot[i] = 0.0;
  __syncthreads();

  for (;;) {
    if (blockIdx.x >= n_owners) break;
    if (tid == 0) s_t = atomic_add_u64(next_target, 1ull);
    __syncthreads();
    const unsigned long long ti = s_t;
    if (ti >= n_targets) break;
    n_search++;
    DN_AT(1, ti)
    const int32_t t_old = target_list[ti];
    const int32_t t = old2new[t_old];
    uint32_t nf = 0, which = 0;
    if (tid == 0) {
      // the record buffer is full: not run, listed for the host's next pass (see the LDS tier)
      s_job[1] = (__hip_atomic_load(O.out_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != ~0ull) ? 1u : 0u;
      if (s_job[1]) {
        O.overflow_list[atomic_add_u64(O.overflow_count, 1ull)] = -(t_old + 1);
        atomic_add_u64(done_targets, 1ull);
      }
    }
    __syncthreads();
    if (s_job[1]) {
      __syncthreads();
      continue;
    }
    if (tid == 0) {
      s_pcount = 0;
      s_tcount = 0;
      s_giveup = 0;
      const bool hot_t = (uint32_t)t < hot_n;
      if (in_rp[t + 1] == in_rp[t]) {  // Backward_Search.java:46-49: reserve = {t: 1.0}
        if (hot_t) hrsv[t] = 1.0;
        else dn_store(&W.rsv[t], 1.0);
        dn_store(&W.plist[0], t);
        s_pcount = 1;
      } else {
        if (hot_t) {
          hres[t] = 1.0;  // :54-56; the target is pushed unconditionally first
        } else {
          dn_store(&W.res[t], 1.0);
          dn_store(&W.touched[0], t);
          s_tcount = 1;
        }
        dn_store(&W.fr[0][0], t);
        nf = 1;
      }
      s_job[0] = nf;
    }
    __syncthreads();
    nf = s_job[0];

    while (nf > 0) {
      // ---- every frontier node gives up its residue (:58-67,72; a node is in a level's frontier at most once) and
      // the level's edge space is laid out, a tile of kDnStage frontier entries at a time: pending contribution, first
      // in-edge, exclusive prefix of the in-degrees.  A level of one tile stays in LDS; a larger one (or one whose
      // edges are worth sharing) is written to the workspace, with the frontier position every chunk of edges starts in.
      const int32_t* cur = W.fr[which];
      const bool one_tile = nf <= (uint32_t)kDnStage;
      DN_AT(2, nf)
      if (tid == 0) {
        s_carry = 0;
        s_nnext = 0;
      }
      __syncthreads();
      for (uint32_t tile = 0; tile < nf; tile += kDnThreads) {
        const uint32_t i = tile + tid;
        uint32_t d = 0, b = 0;
        double pc = 0.0;
        if (i < nf) {
          const int32_t v = dn_load(&cur[i]);
          double rc;
          if ((uint32_t)v < hot_n) {  // (a node is in a level's frontier once: plain LDS accesses)
            rc = hres[v];
            hres[v] = 0.0;
            const double r0 = hrsv[v];
            if (r0 == 0.0) {  // first pop: listed like a cold node (emission and clean-up walk the list)
              const uint32_t pp = atomicAdd(&s_pcount, 1u);
              if (pp < cap_f) dn_store(&W.plist[pp], v);
              else s_giveup = 1;
            }
            hrsv[v] = r0 + rc * alpha;
          } else {
            rc = __hip_atomic_exchange(&W.res[v], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const double r0 = dn_load(&W.rsv[v]);
            if (r0 == 0.0) {  // first pop of this node (alpha * rc > 0 ever after)
              const uint32_t pp = atomicAdd(&s_pcount, 1u);
              if (pp < cap_f) dn_store(&W.plist[pp], v);
              else s_giveup = 1;
            }
            dn_store(&W.rsv[v], r0 + rc * alpha);
          }
          pc = (1.0 - alpha) * rc;
          b = in_rp[v];
          d = in_rp[v + 1] - b;
        }
        uint32_t tile_tot = 0;
        const uint32_t ex = s_carry + block_excl_scan_n<uint32_t, kDnWaves>(d, s_scan, &tile_tot);
        if (one_tile) {  // (tile == 0)
          if (i < nf) {
            L.eoff[tid] = ex;
            L.row[tid] = b;
            L.c[tid] = pc;
          }
        } else if (i < nf) {
          dn_store(&W.pend[i], pc);
          dn_store(&W.frow[i], b);
          dn_store(&W.eoff[i], ex);
          if (d) {  // chunks whose first edge lies in this node's range start here
            const uint32_t c_lo = (uint32_t)(((unsigned long long)ex + chunk - 1) / chunk);
            const uint32_t c_hi = (uint32_t)(((unsigned long long)ex + d - 1) / chunk);
            for (uint32_t c = c_lo; c <= c_hi; ++c) dn_store(&W.cstart[c], i);
          }
        }
        __syncthreads();
        if (tid == 0) s_carry += tile_tot;
        __syncthreads();
      }
      const uint32_t E = s_carry;
      const uint32_t n_chunks = (uint32_t)(((unsigned long long)E + chunk - 1) / chunk);
      const bool post = !one_tile || (share && n_chunks >= kDnShareMin);
      pops += (tid == 0) ? nf : 0;
      edges += (tid == 0) ? E : 0;
      if (!post) {
        // ---- the level stays here: edges straight from the LDS tile, appends counted in LDS
        if (tid == 0) L.eoff[nf] = E;
        __syncthreads();
        DN_TICK(0)
        DN_AT(5, E)
        dn_edge_range<false>(W, L, nf, 0u, E, W.fr[which ^ 1], S_local, in_rec, rmax, cap_t, cap_f, hres, hot_n);
        dn_barrier_global();  // the appends have arrived before the next level reads them
        DN_TICK(1)
        if (tid == 0) {
          s_gaveup = (s_giveup || dn_load(abort_word)) ? 1u : 0u;
          s_job[0] = s_gaveup ? 0u : s_nnext;
        }
      } else {
        // ---- the level goes through the workspace: a one-tile level is written out first
        if (one_tile && (uint32_t)tid < nf) {
          const uint32_t ex = L.eoff[tid], nx = (uint32_t)tid + 1 < nf ? L.eoff[tid + 1] : E;
          dn_store(&W.pend[tid], L.c[tid]);
          dn_store(&W.frow[tid], L.row[tid]);
          dn_store(&W.eoff[tid], ex);
          if (nx > ex) {
            const uint32_t c_lo = (uint32_t)(((unsigned long long)ex + chunk - 1) / chunk);
            const uint32_t c_hi = (uint32_t)(((unsigned long long)nx - 1) / chunk);
            for (uint32_t c = c_lo; c <= c_hi; ++c) dn_store(&W.cstart[c], (uint32_t)tid);
          }
        }
        dn_hot_spill(W, hres, hot_n);  // helpers reach the hot residues through the global vector only
        if (tid == 0) {
          dn_store(&W.eoff[nf], E);
          // the appends of a posted level are counted on the board entry (helpers add to them)
          dn_reset(&B->tcount, s_tcount);
          dn_reset(&B->nnext, 0u);
          dn_reset(&B->giveup, s_giveup);
        }
        dn_barrier_global();  // the level's layout has arrived in memory before it is posted or read
        DN_TICK(0)
        // post the level; take chunks of it like any helper; go on when all of them are done
        if (tid == 0) {
          dn_store(&B->n_chunks, n_chunks);
          dn_store(&B->nf, nf);
          dn_store(&B->which, which);
          dn_reset(&B->done, 0u);  // (includes the release fence for the stores above)
          seq += 1;                // odd: open
          dn_reset(&B->next, (unsigned long long)seq << 32);
          if (share) atomic_add_u64(n_open, 1ull);  // what idle workgroups poll
          s_job[0] = (uint32_t)atomic_add_u64(&B->next, 1ull);  // first chunk (only helpers need the compare-and-swap)
        }
        __syncthreads();
        // NOTE on the shape of this loop (and of every loop in this kernel that holds barriers): a block that only
        // lane 0 executes must never be the last thing before the back edge while another such block opens the loop -
        // the compiler then threads the two together, lanes other than 0 loop back to the barrier on a path of their
        // own, and the waves pass it before lane 0 has written what they are about to read (seen in the ISA of an
        // earlier form of this loop: every wave re-read chunk 0 for ever).  Lane-0 blocks are followed by a barrier.
        for (uint32_t c = s_job[0]; c < n_chunks; c = s_job[0]) {
          __syncthreads();  // s_job[0] has been read by every wave
          DN_AT(3, c)
          dn_chunk(W, B, which, nf, E, c, in_rec, rmax, D, L, hot_n);
          if (tid == 0) {  // (dn_chunk ended with every wave's stores acknowledged)
            atomicAdd(&B->done, 1u);
            s_job[0] = (uint32_t)atomic_add_u64(&B->next, 1ull);
          }
          __syncthreads();
        }
        __syncthreads();
        DN_TICK(1)
        if (tid == 0) {
          // bounded: a wait that outlasts kDnWaitTicks (a bug, or a helper lost to a fault) raises the launch's abort
          // word, on which every loop of every workgroup ends; the host turns it into an error
          DN_AT(4, n_chunks)
          const unsigned long long t_wait = wall_clock64();
          while (dn_load(&B->done) < n_chunks && !dn_load(abort_word)) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t_wait > kDnWaitTicks) atomic_add_u64(abort_word, 1ull);
          }
          seq += 1;  // even: closed (a helper's compare-and-swap on the old sequence fails from here on)
          dn_reset(&B->next, (unsigned long long)seq << 32);
          if (share) atomic_add_u64(n_open, ~0ull);
          s_tcount = dn_load(&B->tcount);
          s_giveup = dn_load(&B->giveup);
          s_gaveup = (s_giveup || dn_load(abort_word)) ? 1u : 0u;
          s_job[0] = s_gaveup ? 0u : dn_load(&B->nnext);
        }
        __syncthreads();
        dn_hot_fill(W, hres, hot_n);  // the level is closed: hot residues back into LDS, the vector's head zero again
        dn_barrier_global();
        DN_TICK(2)
      }
      __syncthreads();
      nf = s_job[0];
      which ^= 1;
      __syncthreads();
    }

    DN_AT(6, 0)
    // a list is full (or the launch is being aborted): the search goes to the whole-vector tier
    if (tid == 0) s_gaveup = (s_giveup || dn_load(abort_word)) ? 1u : 0u;
    __syncthreads();
    const bool gave_up = s_gaveup != 0;
    const uint32_t np = s_pcount < cap_f ? s_pcount : cap_f;
    // ---- emit entries >= threshold (Base_Whole_Graph.java:80-88): only popped nodes hold a reserve
    bool retry = gave_up;
    // Round 5: a search of up to kDnEmitRegs * 1024 pops (all but the hubs') reads its reserves ONCE: they stay in
    // registers between the count and the emission, and the cleared value goes back in the same walk - the count
    // pass, the emission pass and the clean-up's pass over the pop list were three gathers of the same scattered
    // entries (emission 6 % + clean-up 15 % of the tier's workgroup time, profiles/r05_apbs_summary.md).
    constexpr int kDnEmitRegs = 4;
    bool rsv_cleared = false;
    if (!gave_up && np <= (uint32_t)kDnEmitRegs * kDnThreads) {
      int32_t pv[kDnEmitRegs];
      double pr[kDnEmitRegs];
      unsigned long long run = 0;
#pragma unroll
      for (int j = 0; j < kDnEmitRegs; ++j) {
        const uint32_t i = (uint32_t)j * kDnThreads + tid;
        pv[j] = i < np ? dn_load(&W.plist[i]) : -1;
      }
#pragma unroll
      for (int j = 0; j < kDnEmitRegs; ++j) {
        pr[j] = 0.0;
        if (pv[j] >= 0) {
          if ((uint32_t)pv[j] < hot_n) {
            pr[j] = hrsv[pv[j]];
            hrsv[pv[j]] = 0.0;
          } else {
            pr[j] = dn_load(&W.rsv[pv[j]]);
          }
        }
        run += (pr[j] > 0.0 && pr[j] >= rmax) ? 1ull : 0ull;
      }
#pragma unroll
      for (int j = 0; j < kDnEmitRegs; ++j)
        if (pv[j] >= 0 && (uint32_t)pv[j] >= hot_n) dn_store(&W.rsv[pv[j]], 0.0);  // (a node is popped-listed once)
      rsv_cleared = true;
      const unsigned long long total = block_sum_u64(run, s_scan64);  // valid in thread 0
      if (tid == 0) {
        s_tot = total;
        s_out_base = total ? atomic_add_u64(O.out_count, total) : 0ull;
      }
      __syncthreads();
      const unsigned long long tot = s_tot;
      if (s_out_base + tot > O.out_cap) {
        retry = true;  // the triple buffer is full: the host drains it and runs this target again
        if (tid == 0) atomicMin(O.out_valid, s_out_base);
      } else if (tot) {
        unsigned long long at = s_out_base;
#pragma unroll
        for (int j = 0; j < kDnEmitRegs; ++j) {
          if ((uint32_t)j * kDnThreads >= np) break;  // (uniform)
          const bool take = pr[j] > 0.0 && pr[j] >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex2 =
              block_excl_scan_n<unsigned long long, kDnWaves>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) O.out_rec[at + ex2] = TripleRec{new2old[pv[j]], t_old, pr[j]};
          at += chunk_total;
        }
      }
    } else if (!gave_up) {
      // (a popped hot id's reserve is in LDS)
      unsigned long long run = 0;
      for (uint32_t c0 = 0; c0 < np; c0 += kDnThreads) {
        const uint32_t i = c0 + tid;
        double r = 0.0;
        if (i < np) {
          const int32_t v = dn_load(&W.plist[i]);
          r = (uint32_t)v < hot_n ? hrsv[v] : dn_load(&W.rsv[v]);
        }
        run += (r > 0.0 && r >= rmax) ? 1ull : 0ull;
      }
      const unsigned long long total = block_sum_u64(run, s_scan64);  // valid in thread 0
      if (tid == 0) {
        s_tot = total;
        s_out_base = total ? atomic_add_u64(O.out_count, total) : 0ull;
      }
      __syncthreads();
      const unsigned long long tot = s_tot;
      if (s_out_base + tot > O.out_cap) {
        retry = true;  // the triple buffer is full: the host drains it and runs this target again
        if (tid == 0) atomicMin(O.out_valid, s_out_base);
      } else {
        unsigned long long at = s_out_base;
        for (uint32_t c0 = 0; c0 < np; c0 += kDnThreads) {
          const uint32_t i = c0 + tid;
          const int32_t v = i < np ? dn_load(&W.plist[i]) : 0;
          double r = 0.0;
          if (i < np) {  // read for the last time: the cleared value goes back in the same walk
            if ((uint32_t)v < hot_n) {
              r = hrsv[v];
              hrsv[v] = 0.0;
            } else {
              r = dn_load(&W.rsv[v]);
              dn_store(&W.rsv[v], 0.0);
            }
          }
          const bool take = r > 0.0 && r >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex2 =
              block_excl_scan_n<unsigned long long, kDnWaves>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) {
            O.out_rec[at + ex2] = TripleRec{new2old[v], t_old, r};
          }
          at += chunk_total;
        }
        rsv_cleared = true;
      }
    }
    if (retry && tid == 0) {  // lists full: +t, triple buffer full: -(t + 1)
      const unsigned long long p = atomic_add_u64(O.overflow_count, 1ull);
      O.overflow_list[p] = gave_up ? t_old : -(t_old + 1);
    }
    __syncthreads();
    DN_TICK(3)
    DN_AT(7, 0)
    // ---- hand the vectors back all-zero (the hot ids' state: LDS; the head of the global vector is zero already)
    for (uint32_t i = tid; i < hot_n; i += kDnThreads) hres[i] = 0.0;  // (what the last level left below the threshold)
    if (!rsv_cleared) {
      for (uint32_t i = tid; i < np; i += kDnThreads) {
        const int32_t v = dn_load(&W.plist[i]);
        if ((uint32_t)v < hot_n) hrsv[v] = 0.0;
        else dn_store(&W.rsv[v], 0.0);
      }
    }
    const uint32_t nt = s_tcount;
    if (nt <= cap_t && s_pcount <= cap_f) {
      // (tried in round 5: eight lanes per listed node zeroing the 64 aligned bytes around it, so that the memory side
      // sees whole 64-byte writes - 545 / 512 ms against 563 / 527 ms for R-MAT 22's tier, 1 668 / 1 643 against 1 646 /
      // 1 618 ms for R-MAT 24's on the same box: no difference beyond the noise; the 8-byte stores stay)
      for (uint32_t i = tid; i < nt; i += kDnThreads) dn_store(&W.res[dn_load(&W.touched[i])], 0.0);
    } else {  // a list overflowed: clear everything
      for (uint32_t i = tid; i < n; i += kDnThreads) {
        dn_store(&W.res[i], 0.0);
        dn_store(&W.rsv[i], 0.0);
      }
      for (uint32_t i = tid; i < hot_n; i += kDnThreads) hrsv[i] = 0.0;
    }
    dn_barrier_global();  // the zeros have arrived before the next search's atomics can meet them
    if (tid == 0) atomic_add_u64(done_targets, 1ull);
    DN_TICK(4)
    __syncthreads();  // (lane-0 blocks never sit next to a back edge: see the note at the chunk loop)
    // ---- between two searches of its own the workgroup holds no state: in a short pass (launch_apbs) levels that are
    // open on other workgroups' boards come first.  The targets are taken largest first, so the open levels belong to
    // the searches the launch will end with; pushed early they are not what everybody waits for at the end.
    if (share && help_between) {
      DN_AT(8, 2)
      while (dn_help_once(H, L, board, ws_base, D, in_rec, rmax, hot_n, n_owners, n_open, done_targets, n_targets,
                          abort_word) != 0) {
      }
      DN_TICK(5)
    }
  }

  // ---- out of targets: help with posted levels until every target of the launch is finished
  if (share) {
    const unsigned long long t_idle = wall_clock64();
    DN_AT(8, 0)
    for (;;) {
      if (tid == 0 && wall_clock64() - t_idle > 4 * kDnWaitTicks) atomic_add_u64(abort_word, 1ull);
      const int r = dn_help_once(H, L, board, ws_base, D, in_rec, rmax, hot_n, n_owners, n_open, done_targets, n_targets,
                                 abort_word);
      if (r != 0) continue;  // pushed a chunk, or lost one to somebody faster: look again at once
      if (H.done) break;
      __syncthreads();  // (H.done has been read by every wave before lane 0 writes it again)
      for (int k = 0; k < 4; ++k) __builtin_amdgcn_s_sleep(127);  // ~15 us between polls
    }
    DN_TICK(5)
  }
  DN_AT(10, 0)
#undef DN_TICK
#undef DN_AT
  const unsigned long long ps = block_sum_u64(pops, s_scan64);
  const unsigned long long es = block_sum_u64(edges, s_scan64);
  if (tid == 0) {
    if (ps) atomic_add_u64(O.stat_pops, ps);
    if (es) atomic_add_u64(O.stat_edges, es);
    if (dbg) {
      unsigned long long* d = dbg + (size_t)blockIdx.x * 12;
      d[0] = n_search;
      d[1] = edges;
      for (int i = 0; i < 6; ++i) d[2 + i] = tk[i];
      d[8] = wall_clock64();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sharded All-Pair: index entries as 16-byte records, partitioned by the owner of their source
// ------------------------------------------------------------------------------------------------
constexpr int kMaxWorld = 64;
constexpr int kPartTile = 2048;  // records per workgroup pass

// pass 0 (out == nullptr): counts per owner; pass 1: records land in their owner's segment.  One global atomic
// per owner and tile (LDS counts first), so 10^7-10^8 records do not pile onto `world` addresses.
__global__ __launch_bounds__(256) void k_owner_partition(const TripleRec* __restrict__ rec, unsigned long long count,
                                                          uint32_t base, uint32_t rem, int world,
                                                          unsigned long long* __restrict__ cursors,
                                                          TripleRec* __restrict__ out) {
  __shared__ uint32_t s_cnt[kMaxWorld];
  __shared__ unsigned long long s_base[kMaxWorld];
  const unsigned long long n_tiles = (count + kPartTile - 1) / kPartTile;
  for (unsigned long long tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    if (threadIdx.x < kMaxWorld) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t own[kPartTile / 256], pos[kPartTile / 256];
#pragma unroll
    for (int j = 0; j < kPartTile / 256; ++j) {
      const unsigned long long i = tl * kPartTile + (unsigned long long)j * 256 + threadIdx.x;
      own[j] = 0xFFFFFFFFu;
      if (i < count) {
        own[j] = owner_of((uint32_t)rec[i].v, base, rem);
        pos[j] = atomicAdd(&s_cnt[own[j]], 1u);
      }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)world && s_cnt[threadIdx.x])
      s_base[threadIdx.x] = atomic_add_u64(&cursors[threadIdx.x], (unsigned long long)s_cnt[threadIdx.x]);
    __syncthreads();
    if (out) {
#pragma unroll
      for (int j = 0; j < kPartTile / 256; ++j) {
        const unsigned long long i = tl * kPartTile + (unsigned long long)j * 256 + threadIdx.x;
        if (own[j] != 0xFFFFFFFFu) out[s_base[own[j]] + pos[j]] = rec[i];
      }
    }
    __syncthreads();
  }
}

// cursors: `world` counters, zero before pass 0 (counts), holding the segment starts before pass 1
int launch_owner_partition(pprhip_graph* g, const TripleRec* rec, unsigned long long count, int world,
                           unsigned long long* cursors, TripleRec* out) {
  if (!count) return PPRHIP_OK;
  if (world > kMaxWorld) {
    set_error("owner partition: at most %d ranks", kMaxWorld);
    return PPRHIP_ERR_INVALID;
  }
  const uint32_t grid = (uint32_t)std::min<unsigned long long>((count + kPartTile - 1) / kPartTile, 2048ull);
  k_owner_partition<<<dim3(grid), dim3(256), 0, g->stream>>>(rec, count, g->n / (uint32_t)world, g->n % (uint32_t)world,
                                                             world, cursors, out);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// ------------------------------------------------------------------------------------------------
// the entries of ONE whole-vector backward search (the handle's own residue / reserve vectors): reserve >= threshold
// -> 16-byte records, one reservation per wave
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_emit_reserve(const double* __restrict__ reserve, uint32_t n, double rmax,
                                                       const int32_t* __restrict__ new2old, int32_t t_old,
                                                       TripleRec* __restrict__ out, unsigned long long cap,
                                                       unsigned long long* __restrict__ count) {
  const int lane = lane_id();
  const uint32_t stride = gridDim.x * 256u;
  for (uint32_t v0 = blockIdx.x * 256u; v0 < n; v0 += stride) {  // (wave-uniform trip count)
    const uint32_t v = v0 + threadIdx.x;
    const double r = v < n ? reserve[v] : 0.0;
    const bool take = r > 0.0 && r >= rmax;  // Base_Whole_Graph.java:83 (only entries of the reserve map exist)
    const unsigned long long mask = __ballot(take);
    if (!mask) continue;
    unsigned long long base = 0;
    if (lane == 0) base = atomic_add_u64(count, (unsigned long long)__popcll(mask));
    base = __shfl(base, 0);
    if (take) {
      const unsigned long long pos = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
      if (pos < cap) out[pos] = TripleRec{new2old ? new2old[v] : (int32_t)v, t_old, r};
    }
  }
}

int launch_emit_reserve(pprhip_graph* g, const double* reserve, uint32_t n, double rmax, int32_t t_old, TripleRec* out,
                        unsigned long long cap, unsigned long long* count) {
  if (!n) return PPRHIP_OK;
  const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)n + 255) / 256, 4096);
  k_emit_reserve<<<dim3(grid), dim3(256), 0, g->stream>>>(reserve, n, rmax, g->relabeled ? g->new2old : nullptr, t_old, out, cap,
                                                          count);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int init_kernels_apbs() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_lds<kApLdsCap, kApFront, 256>)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_lds<kApSmallCap, kApSmallFront, kApSmallThreads>)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_split)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_dense)));
  PPRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_apbs_dense), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)(2 * sizeof(double) * kDnHotMax)));
  return PPRHIP_OK;
}

int launch_build_in_rec(pprhip_graph* g, void* rec) {
  if (!g->m) return PPRHIP_OK;
  const uint32_t grid = (uint32_t)std::min<unsigned long long>((g->m + 255) / 256, 8192ull);
  k_build_in_rec<<<dim3(grid), dim3(256), 0, g->stream>>>(g->in_ci, g->out_ext, (unsigned long long)g->m, (InRec*)rec);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_apbs(pprhip_graph* g, bool dense_tier, const int32_t* d_targets, uint32_t t_begin, uint32_t n_targets,
                double alpha, double rmax, ApbsBuffers& b) {
  const ApOut O{b.out_rec, b.out_cap, b.out_count, b.out_valid, b.overflow, b.overflow_count,
                b.stat_pops, b.stat_edges};
  const InRec* rec = (const InRec*)g->in_rec;
  if (dense_tier) {
    if (!d_targets || !b.ws || !b.ws_blocks || !b.board || !b.done_targets || !b.chunk) {
      set_error("All-Pair dense tier: no target list or workspace");
      return PPRHIP_ERR_STATE;
    }
    const char* sh = hook_env("PPRHIP_APBS_SHARE");  // developer switch: 0 = every search stays on its own workgroup
    const int share = (sh && sh[0] == '0') ? 0 : 1;
    const uint32_t owners = std::min<uint32_t>(std::min<uint32_t>(b.ws_blocks, (uint32_t)kDnThreads), std::max(1u, n_targets));
    // helpers beyond the owners (b.helpers: workgroups the launch may use in all) only make sense when levels are shared
    const uint32_t grid = share ? std::max(owners, std::min<uint32_t>(b.helpers, (uint32_t)kDnThreads)) : owners;
    // ids whose residue / reserve live in the owner's LDS (dn_edge_range): 8 K by default (128 KB), at most a quarter
    // of the graph so that small graphs still run both paths (PPRHIP_APBS_HOT: developer / test switch, 0 = none)
    const char* he = hook_env("PPRHIP_APBS_HOT");
    uint32_t hot_n = he ? (uint32_t)std::max(0, atoi(he)) : (uint32_t)kDnHotDefault;
    hot_n = std::min<uint32_t>(std::min<uint32_t>(hot_n, (uint32_t)kDnHotMax), g->n / 4);
    // Workgroups also help between two searches of their own when the pass is short (below 2^17 targets): there the
    // largest searches are what the launch ends with, and help that comes early shortens the end - R-MAT 22, tier 2 of
    // 2^18 / 2^19 targets of the range: 49.5 -> 44.3 ms / 79.1 -> 74.9 ms.  A long pass has no idle end to shorten
    // (17 workgroup-ms of 138 000 at 617 K searches) and pays for the looks at the boards and the races for chunks:
    // 139 -> 144 ms at 2^20 targets, 540 -> 586 ms at all 2^22.  (PPRHIP_APBS_HELP_BETWEEN=0|1: test switch.)
    const char* hb = hook_env("PPRHIP_APBS_HELP_BETWEEN");
    const int help_between = hb ? (hb[0] != '0') : (n_targets < (1u << 17));
    k_apbs_dense<<<dim3(grid), dim3(kDnThreads), 2 * sizeof(double) * (size_t)hot_n, g->stream>>>(
        d_targets, n_targets, b.next_target, g->in_rp, rec, g->old2new, g->new2old, alpha, rmax, O, b.ws, (DnBoard*)b.board,
        b.done_targets, b.done_targets + 1, b.done_targets + 2, dims_of(g->n, (unsigned long long)g->m, b.cap_t, b.cap_f, b.chunk), share,
        owners, hot_n, help_between, b.dbg);
  } else if (!d_targets && b.list0 && b.list1) {
    // a range, in three steps on the device: trivial targets and the list of the others; the small table; the large
    // table for what the small one gave up (cells: +6 the list's length, +7 the small table's target cursor, +11 the
    // length of its give-up list; b.next_target is the large table's cursor)
    unsigned long long* cells = b.next_target;
    const uint32_t sgrid = (uint32_t)std::min<uint64_t>(((uint64_t)n_targets + 255) / 256, 2048);
    // (PPRHIP_APBS_DEG=big,dense: the in-degrees from which a search starts in the large table / in the dense tier -
    // developer and test switch; "0,0" sends everything through both tables as before round 4)
    // (R-MAT 22, searches of all 4.19 M targets: 3,9 649 ms / 4,12 655 / 6,16 667 / 4,24 665 / all through both tables
    // 783; R-MAT 24: 4,12 1 894 ms / 3,9 2 019)
    uint32_t deg_big = 4, deg_dense = 12;
    if (const char* de = hook_env("PPRHIP_APBS_DEG")) {
      unsigned a = 0, c = 0;
      if (sscanf(de, "%u,%u", &a, &c) == 2) {
        deg_big = a ? a : 0xFFFFFFFFu;
        deg_dense = c ? std::max(c, a) : 0xFFFFFFFFu;
      }
    }
    k_apbs_split<<<dim3(sgrid), dim3(256), 0, g->stream>>>(t_begin, n_targets, g->in_rp, g->old2new, rmax, O, b.list0, cells + 6,
                                                           b.list1, cells + 11, deg_big, deg_dense);
    PPRHIP_CHECK_HIP(hipGetLastError());
    ApOut O0 = O;  // the small table's give-ups (and retries) go to the second list
    O0.overflow_list = b.list1;
    O0.overflow_count = cells + 11;
    const uint32_t grid0 = std::min<uint32_t>((uint32_t)g->n_cus * 8u, std::max(1u, n_targets));
    k_apbs_lds<kApSmallCap, kApSmallFront, kApSmallThreads><<<dim3(grid0), dim3(kApSmallThreads), 0, g->stream>>>(
        b.list0, 0u, 0u, cells + 6, cells + 7, g->in_rp, rec, g->old2new, g->new2old, alpha, rmax, O0);
    PPRHIP_CHECK_HIP(hipGetLastError());
    const uint32_t grid1 = std::min<uint32_t>((uint32_t)g->n_cus * 2u, std::max(1u, n_targets));
    k_apbs_lds<kApLdsCap, kApFront, 256><<<dim3(grid1), dim3(256), 0, g->stream>>>(
        b.list1, 0u, 0u, cells + 11, b.next_target, g->in_rp, rec, g->old2new, g->new2old, alpha, rmax, O);
  } else {
    const uint32_t grid = std::min<uint32_t>((uint32_t)g->n_cus * 2u, std::max(1u, n_targets));
    k_apbs_lds<kApLdsCap, kApFront, 256><<<dim3(grid), dim3(256), 0, g->stream>>>(d_targets, t_begin, n_targets, nullptr, b.next_target,
                                                                             g->in_rp, rec, g->old2new, g->new2old, alpha, rmax, O);
  }
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
