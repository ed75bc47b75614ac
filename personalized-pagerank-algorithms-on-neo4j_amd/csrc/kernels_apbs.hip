// kernels_apbs.hip — All-Pair-Backward-Search: many targets in flight (gfx950).
//
// Base_Whole_Graph.preprocessing (Base_Whole_Graph.java:58-164) runs one backward search
// (Backward_Search.java:38-100) per target; the searches are independent.  A persistent workgroup takes one
// target at a time and keeps that target's whole state (residue, reserve, frontier) to itself:
//
//   tier 1  hash table of 2048 entries in LDS (ds_cmpst / ds_add_rtn_f64), two workgroups per CU: nine searches in
//           ten touch fewer than 1536 nodes (R-MAT 22, threshold 1e-3: 89 % of the targets, 8 % of the edges);
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

constexpr int kApLdsCap = 2048;
constexpr int kApFront = 512;

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
__device__ __forceinline__ uint32_t ap_slot(const ApTable& T, int32_t u, uint32_t* used_count, uint32_t limit,
                                            uint32_t* overflow) {
  constexpr uint32_t mask = kApLdsCap - 1;
  uint32_t s = ((uint32_t)u * 2654435761u) >> 7 & mask;
  for (uint32_t probes = 0; probes < (uint32_t)kApLdsCap; ++probes) {
    const int32_t k = T.keys[s];
    if (k == u) return s;
    if (k == -1) {
      const int32_t prev = atomicCAS(&T.keys[s], -1, u);  // ds_cmpst_rtn_b32
      if (prev == -1) {
        const uint32_t idx = atomicAdd(used_count, 1u);
        if (idx < (uint32_t)kApLdsCap) T.used[idx] = (uint16_t)s;
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
  int32_t* out_v;
  int32_t* out_t;
  double* out_p;
  unsigned long long out_cap;
  unsigned long long* out_count;
  unsigned long long* out_valid;
  int32_t* overflow_list;
  unsigned long long* overflow_count;
  unsigned long long* stat_pops;
  unsigned long long* stat_edges;
};

__global__ __launch_bounds__(256) void k_apbs_lds(const int32_t* __restrict__ target_list, uint32_t t_begin,
                                                   uint32_t n_targets, unsigned long long* next_target,
                                                   const uint32_t* __restrict__ in_rp, const InRec* __restrict__ in_rec,
                                                   const int32_t* __restrict__ old2new, const int32_t* __restrict__ new2old,
                                                   double alpha, double rmax, ApOut O) {
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
  __shared__ uint32_t s_scan[4];
  __shared__ unsigned long long s_scan64[4];
  __shared__ uint32_t s_used_count, s_nnext, s_overflow;
  __shared__ unsigned long long s_t, s_out_base, s_tot;
  const int tid = threadIdx.x;

  ApTable T{s_keys, s_res, s_rsv, s_pend, s_used, s_cur, s_nxt};
  const uint32_t limit = kApLdsCap - kApLdsCap / 4;  // give up at 75 % load
  for (uint32_t i = tid; i < (uint32_t)kApLdsCap; i += 256) {
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
    const int32_t t_old = target_list ? target_list[ti] : (int32_t)(t_begin + ti);
    const int32_t t = old2new[t_old];
    if (tid == 0) {
      s_used_count = 0;
      s_overflow = 0;
    }
    __syncthreads();
    uint32_t nf = 0;
    if (in_rp[t + 1] == in_rp[t]) {  // Backward_Search.java:46-49: reserve = {t: 1.0}
      if (tid == 0) {
        const uint32_t s = ap_slot(T, t, &s_used_count, limit, &s_overflow);
        T.rsv[s] = 1.0;
      }
    } else {
      if (tid == 0) {
        const uint32_t s = ap_slot(T, t, &s_used_count, limit, &s_overflow);
        T.res[s] = 1.0;  // :54-56; the target is pushed unconditionally first
        T.cur[0] = (uint16_t)s;
      }
      nf = 1;
    }
    __syncthreads();

    while (nf > 0 && !s_overflow) {
      // ---- every frontier node gives up its residue (:58-67,72)
      for (uint32_t i = tid; i < nf; i += 256) {
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
          const uint32_t i0 = tid, i1 = tid + 256;
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
        // exclusive prefix of the degrees over the sub-batch (two elements per thread: i, i + 256)
        uint32_t tot0 = 0, tot1 = 0;
        const uint32_t e0 = block_excl_scan_256<uint32_t>(d0, s_scan, &tot0);
        const uint32_t e1 = block_excl_scan_256<uint32_t>(d1, s_scan, &tot1);
        if ((uint32_t)tid < cnt) f_off[tid] = e0;
        if ((uint32_t)tid + 256 < cnt) f_off[tid + 256] = tot0 + e1;
        if (tid == 0) f_off[cnt] = tot0 + tot1;
        __syncthreads();
        const uint32_t E = f_off[cnt];
        edges += (tid == 0) ? E : 0;
        // four edges per thread in flight: edge records first, then table updates
        for (uint32_t base = tid; base < E; base += 1024) {
          InRec rc4[4];
          double cc[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t e = base + 256u * q;
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
            const uint32_t s = ap_slot(T, rc4[q].u, &s_used_count, limit, &s_overflow);
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
      for (uint32_t c0 = 0; c0 < used; c0 += 256) {  // count first, one reservation per target
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
        for (uint32_t c0 = 0; c0 < used; c0 += 256) {
          const uint32_t i = c0 + tid;
          const uint32_t s = i < used ? T.used[i] : 0u;
          const bool take = i < used && T.rsv[s] > 0.0 && T.rsv[s] >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex =
              block_excl_scan_256<unsigned long long>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) {
            O.out_v[at + ex] = new2old[T.keys[s]];
            O.out_t[at + ex] = t_old;
            O.out_p[at + ex] = T.rsv[s];
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
      for (uint32_t i = tid; i < (uint32_t)kApLdsCap; i += 256) {
        T.keys[i] = -1;
        T.res[i] = 0.0;
        T.rsv[i] = 0.0;
      }
    } else {
      for (uint32_t i = tid; i < used; i += 256) {
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

// ------------------------------------------------------------------------------------------------
// tier 2: one target's state in dense per-workgroup vectors
// ------------------------------------------------------------------------------------------------
// Workspace of one workgroup (HBM, all-zero between searches): residue[n], reserve[n], and the lists a search keeps:
// the nodes whose residue it has made non-zero (for the clean-up), the current / next frontier with the frontier's
// pending contributions, and the nodes it has popped at least once (the only ones that can hold a reserve).
constexpr int kDnThreads = 1024;
constexpr int kDnWaves = kDnThreads / 64;
constexpr int kDnFront = 1024;  // frontier nodes per sub-batch: one per thread
constexpr int kDnIlp = 4;       // edges a thread keeps in flight

struct DenseWs {
  double* res;
  double* rsv;
  double* pend;      // [cap_f] pending contribution of the current frontier, by frontier position
  int32_t* touched;  // [cap_t]
  int32_t* cur;      // [cap_f]
  int32_t* nxt;      // [cap_f]
  int32_t* plist;    // [cap_f] nodes popped for the first time, in pop order
};

__host__ __device__ inline size_t dense_ws_bytes(uint32_t n, uint32_t cap_t, uint32_t cap_f) {
  return ((size_t)8 * (2 * (size_t)n + cap_f) + (size_t)4 * ((size_t)cap_t + 3 * (size_t)cap_f) + 255) & ~(size_t)255;
}
size_t apbs_dense_bytes(uint32_t n, uint32_t cap_t, uint32_t cap_f) { return dense_ws_bytes(n, cap_t, cap_f); }

// The residue vector is only ever touched by fp64 atomics (which execute at the memory side and keep nothing in L2)
// and by accesses that bypass the caches the same way; mixing in cached loads or stores could pair an atomic with a
// stale line.
__device__ __forceinline__ double dn_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void dn_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(kDnThreads) void k_apbs_dense(const int32_t* __restrict__ target_list, uint32_t n_targets,
                                                            unsigned long long* next_target,
                                                            const uint32_t* __restrict__ in_rp,
                                                            const InRec* __restrict__ in_rec,
                                                            const int32_t* __restrict__ old2new,
                                                            const int32_t* __restrict__ new2old, double alpha,
                                                            double rmax, ApOut O, char* ws_base, uint32_t n,
                                                            uint32_t cap_t, uint32_t cap_f,
                                                            unsigned long long* __restrict__ dbg) {
  // dbg (developer switch PPRHIP_APBS_DEBUG, else nullptr): per workgroup {searches, edges, ticks of the 100 MHz
  // clock spent in pops, scans, edge loops, emission, clean-up, and the tick at which the workgroup ended}
  __shared__ uint32_t f_row[kDnFront];
  __shared__ uint32_t f_off[kDnFront + 1];
  __shared__ double f_c[kDnFront];
  __shared__ uint32_t s_scan[kDnWaves];
  __shared__ unsigned long long s_scan64[kDnWaves];
  __shared__ uint32_t s_tcount, s_nnext, s_pcount, s_giveup;
  __shared__ unsigned long long s_t, s_out_base, s_tot;
  const int tid = threadIdx.x, lane = lane_id();

  DenseWs W;
  {
    char* base = ws_base + (size_t)blockIdx.x * dense_ws_bytes(n, cap_t, cap_f);
    W.res = (double*)base;
    W.rsv = W.res + n;
    W.pend = W.rsv + n;
    W.touched = (int32_t*)(W.pend + cap_f);
    W.cur = W.touched + cap_t;
    W.nxt = W.cur + cap_f;
    W.plist = W.nxt + cap_f;
  }
  unsigned long long pops = 0, edges = 0;
  unsigned long long tk[5] = {0, 0, 0, 0, 0}, t_mark = dbg ? wall_clock64() : 0ull, n_search = 0;
#define DN_TICK(i)                                   \
  if (dbg && tid == 0) {                             \
    const unsigned long long now_ = wall_clock64();  \
    tk[i] += now_ - t_mark;                          \
    t_mark = now_;                                   \
  }

  for (;;) {
    if (tid == 0) s_t = atomic_add_u64(next_target, 1ull);
    __syncthreads();
    const unsigned long long ti = s_t;
    if (ti >= n_targets) break;
    n_search++;
    const int32_t t_old = target_list[ti];
    const int32_t t = old2new[t_old];
    uint32_t nf = 0;
    if (tid == 0) {
      s_tcount = 0;
      s_pcount = 0;
      s_giveup = 0;
      if (in_rp[t + 1] == in_rp[t]) {  // Backward_Search.java:46-49: reserve = {t: 1.0}
        dn_store(&W.rsv[t], 1.0);
        W.plist[0] = t;
        s_pcount = 1;
      } else {
        dn_store(&W.res[t], 1.0);  // :54-56; the target is pushed unconditionally first
        W.touched[0] = t;
        s_tcount = 1;
        W.cur[0] = t;
      }
    }
    __syncthreads();
    nf = s_tcount;  // 1 when the target has in-edges

    while (nf > 0) {
      // ---- every frontier node gives up its residue (:58-67,72); a node is in a level's frontier at most once
      for (uint32_t i = tid; i < nf; i += kDnThreads) {
        const int32_t v = W.cur[i];
        const double rc = __hip_atomic_exchange(&W.res[v], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double r0 = dn_load(&W.rsv[v]);
        if (r0 == 0.0) {  // first pop of this node (alpha * rc > 0 ever after)
          const uint32_t pp = atomicAdd(&s_pcount, 1u);
          if (pp < cap_f) W.plist[pp] = v;
          else s_giveup = 1;
        }
        dn_store(&W.rsv[v], r0 + rc * alpha);
        W.pend[i] = (1.0 - alpha) * rc;
      }
      if (tid == 0) s_nnext = 0;
      pops += (tid == 0) ? nf : 0;
      __syncthreads();
      DN_TICK(0)
      // ---- in-edges of the frontier, kDnFront frontier nodes at a time
      for (uint32_t fb = 0; fb < nf; fb += kDnFront) {
        const uint32_t cnt = nf - fb < (uint32_t)kDnFront ? nf - fb : (uint32_t)kDnFront;
        uint32_t d = 0;
        if ((uint32_t)tid < cnt) {
          const int32_t v = W.cur[fb + tid];
          const uint32_t b = in_rp[v];
          d = in_rp[v + 1] - b;
          f_row[tid] = b;
          f_c[tid] = W.pend[fb + tid];
        }
        uint32_t E = 0;
        const uint32_t ex = block_excl_scan_n<uint32_t, kDnWaves>(d, s_scan, &E);
        if ((uint32_t)tid < cnt) f_off[tid] = ex;
        if (tid == 0) f_off[cnt] = E;
        __syncthreads();
        edges += (tid == 0) ? E : 0;
        DN_TICK(1)
        // wave-uniform trip count (the appends below are wave-aggregated)
        for (uint32_t base0 = 0; base0 < E; base0 += kDnThreads * kDnIlp) {
          InRec rc4[kDnIlp];
          double add[kDnIlp], old[kDnIlp];
#pragma unroll
          for (int q = 0; q < kDnIlp; ++q) {
            const uint32_t e = base0 + (uint32_t)kDnThreads * q + tid;
            rc4[q] = InRec{-1, 1u};
            add[q] = 0.0;
            if (e < E) {
              uint32_t lo = 0, hi = cnt;  // last frontier entry whose edge range starts at or before e
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (f_off[mid] <= e) lo = mid + 1; else hi = mid;
              }
              const uint32_t i = lo - 1;
              rc4[q] = load_rec(in_rec, (size_t)f_row[i] + (e - f_off[i]));
              add[q] = f_c[i];
            }
          }
#pragma unroll
          for (int q = 0; q < kDnIlp; ++q) {
            add[q] = add[q] / (double)rc4[q].dout;  // :84-85, every edge's quotient rounds on its own
            old[q] = 0.0;
            if (rc4[q].u >= 0) old[q] = atomic_add_ret(&W.res[rc4[q].u], add[q]);
          }
#pragma unroll
          for (int q = 0; q < kDnIlp; ++q) {
            const bool on = rc4[q].u >= 0;
            // a residue leaves zero: remember the node for the clean-up (a node popped in between is listed twice,
            // which only clears it twice)
            const bool first = on && old[q] == 0.0;
            const unsigned long long fm = __ballot(first);
            if (fm) {
              uint32_t wb = 0;
              if (lane == __ffsll((long long)fm) - 1) wb = atomicAdd(&s_tcount, (uint32_t)__popcll(fm));
              wb = __shfl(wb, __ffsll((long long)fm) - 1);
              if (first) {
                const uint32_t pos = wb + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
                if (pos < cap_t) W.touched[pos] = rc4[q].u;
              }
            }
            if (on && !(old[q] > rmax) && old[q] + add[q] > rmax) {  // :89 strict, un-normalised; first crossing
              const uint32_t pos = atomicAdd(&s_nnext, 1u);
              if (pos < cap_f) W.nxt[pos] = rc4[q].u;
              else s_giveup = 1;
            }
          }
        }
        __syncthreads();
        DN_TICK(2)
      }
      nf = s_giveup ? 0u : s_nnext;
      int32_t* tmp = W.cur; W.cur = W.nxt; W.nxt = tmp;
      __syncthreads();
    }

    const bool gave_up = s_giveup != 0;  // a list is full: the search goes to the whole-vector tier
    const uint32_t np = s_pcount < cap_f ? s_pcount : cap_f;
    // ---- emit entries >= threshold (Base_Whole_Graph.java:80-88): only popped nodes hold a reserve
    bool retry = gave_up;
    if (!gave_up) {
      unsigned long long run = 0;
      for (uint32_t c0 = 0; c0 < np; c0 += kDnThreads) {
        const uint32_t i = c0 + tid;
        double r = 0.0;
        if (i < np) r = dn_load(&W.rsv[W.plist[i]]);
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
          const int32_t v = i < np ? W.plist[i] : 0;
          const double r = i < np ? dn_load(&W.rsv[v]) : 0.0;
          const bool take = r > 0.0 && r >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex2 =
              block_excl_scan_n<unsigned long long, kDnWaves>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) {
            O.out_v[at + ex2] = new2old[v];
            O.out_t[at + ex2] = t_old;
            O.out_p[at + ex2] = r;
          }
          at += chunk_total;
        }
      }
    }
    if (retry && tid == 0) {  // lists full: +t, triple buffer full: -(t + 1)
      const unsigned long long p = atomic_add_u64(O.overflow_count, 1ull);
      O.overflow_list[p] = gave_up ? t_old : -(t_old + 1);
    }
    __syncthreads();
    DN_TICK(3)
    // ---- hand the vectors back all-zero
    for (uint32_t i = tid; i < np; i += kDnThreads) dn_store(&W.rsv[W.plist[i]], 0.0);
    const uint32_t nt = s_tcount;
    if (nt <= cap_t && s_pcount <= cap_f) {
      for (uint32_t i = tid; i < nt; i += kDnThreads) dn_store(&W.res[W.touched[i]], 0.0);
    } else {  // a list overflowed: clear everything
      for (uint32_t i = tid; i < n; i += kDnThreads) {
        dn_store(&W.res[i], 0.0);
        dn_store(&W.rsv[i], 0.0);
      }
    }
    __syncthreads();
    DN_TICK(4)
  }
#undef DN_TICK
  const unsigned long long ps = block_sum_u64(pops, s_scan64);
  const unsigned long long es = block_sum_u64(edges, s_scan64);
  if (tid == 0) {
    if (ps) atomic_add_u64(O.stat_pops, ps);
    if (es) atomic_add_u64(O.stat_edges, es);
    if (dbg) {
      unsigned long long* d = dbg + (size_t)blockIdx.x * 8;
      d[0] = n_search;
      d[1] = edges;
      for (int i = 0; i < 5; ++i) d[2 + i] = tk[i];
      d[7] = wall_clock64();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sharded All-Pair: index entries as 16-byte records, partitioned by the owner of their source
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_triples(const int32_t* __restrict__ v, const int32_t* __restrict__ t,
                                                       const double* __restrict__ p, unsigned long long count,
                                                       TripleRec* __restrict__ dst) {
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256ull) {
    TripleRec r;
    r.v = v[i];
    r.t = t[i];
    r.p = p[i];
    dst[i] = r;
  }
}

// rank that owns source v when [0, n) is cut into `world` contiguous ranges (the first n % world one longer)
__device__ __forceinline__ uint32_t owner_of(uint32_t v, uint32_t base, uint32_t rem) {
  const uint32_t cut = rem * (base + 1u);
  return v < cut ? v / (base + 1u) : rem + (v - cut) / base;
}

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

int launch_pack_triples(pprhip_graph* g, const int32_t* v, const int32_t* t, const double* p, unsigned long long count,
                        TripleRec* dst) {
  if (!count) return PPRHIP_OK;
  const uint32_t grid = (uint32_t)std::min<unsigned long long>((count + 255) / 256, 4096ull);
  k_pack_triples<<<dim3(grid), dim3(256), 0, g->stream>>>(v, t, p, count, dst);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
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

int init_kernels_apbs() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_lds)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs_dense)));
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
  const ApOut O{b.out_v, b.out_t, b.out_p, b.out_cap, b.out_count, b.out_valid, b.overflow, b.overflow_count,
                b.stat_pops, b.stat_edges};
  const InRec* rec = (const InRec*)g->in_rec;
  if (dense_tier) {
    if (!d_targets || !b.ws || !b.ws_blocks) {
      set_error("All-Pair dense tier: no target list or workspace");
      return PPRHIP_ERR_STATE;
    }
    const uint32_t grid = std::min<uint32_t>(b.ws_blocks, std::max(1u, n_targets));
    k_apbs_dense<<<dim3(grid), dim3(kDnThreads), 0, g->stream>>>(d_targets, n_targets, b.next_target, g->in_rp, rec,
                                                                 g->old2new, g->new2old, alpha, rmax, O, b.ws, g->n,
                                                                 b.cap_t, b.cap_f, b.dbg);
  } else {
    const uint32_t grid = std::min<uint32_t>((uint32_t)g->n_cus * 2u, std::max(1u, n_targets));
    k_apbs_lds<<<dim3(grid), dim3(256), 0, g->stream>>>(d_targets, t_begin, n_targets, b.next_target, g->in_rp, rec,
                                                        g->old2new, g->new2old, alpha, rmax, O);
  }
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
