// kernels_apbs.hip — All-Pair-Backward-Search: many targets in flight (gfx950).
//
// Base_Whole_Graph.preprocessing (Base_Whole_Graph.java:58-164) runs one backward search
// (Backward_Search.java:38-100) per target; the searches are independent and each touches only
// O(1/(alpha*threshold)) nodes.  A persistent workgroup takes one target at a time and keeps that
// target's whole state (residue, reserve, frontier) in a hash table keyed by node id:
//
//   tier 1  table of 2048 entries in LDS (ds_cmpst / ds_add_rtn_f64), two workgroups per CU;
//   tier 2  table of 524288 entries in HBM per workgroup (filled to a quarter), for the targets tier 1 had to give up;
//   tier 3  (host) the engine's whole-vector backward search for the few targets beyond that.
//
// Levels are frontier-synchronous exactly as in k_sparse_prepare / k_sparse_push: all frontier
// nodes give up their residue first, then their in-edges are expanded edge-parallel (degree
// prefix over sub-batches of 512 frontier nodes), one atomic add per edge, crossing test on
// (old, old + add) with the reference's strict un-normalised threshold (Backward_Search.java:89).
// Entries with reserve >= threshold are appended as (source, target, pi) triples.
#include <algorithm>
#include <type_traits>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

constexpr int kApLdsCap = 2048;
constexpr int kApFront = 512;
constexpr uint32_t kApGlobalLoadDiv = 4;  // the HBM tier hands a search on at cap / 4 nodes
// entries of the HBM tier's lists: the node limit plus what one round of inserts can add before the limit is checked
__host__ __device__ inline uint32_t ap_list_cap(uint32_t g_cap) { return g_cap / kApGlobalLoadDiv + 8192u; }
size_t apbs_table_bytes(uint32_t g_cap) { return (size_t)g_cap * 32 + (size_t)ap_list_cap(g_cap) * 12; }

template <bool G>
struct ApTable {  // one target's state; arrays live in LDS (tier 1) or HBM (tier 2)
  using Idx = typename std::conditional<G, uint32_t, uint16_t>::type;  // slot index (LDS tables have <= 65536 slots)
  int32_t* keys;   // node id or -1
  double* res;     // residue
  double* rsv;     // reserve
  double* pend;    // (1 - alpha) * residue taken at level start, per slot
  Idx* used;       // slots in insertion order
  Idx* cur;        // frontier (slots)
  Idx* nxt;
  uint32_t cap;    // power of two
  uint32_t lcap;   // entries the three lists hold
  uint32_t ds, ks; // element strides of the double fields / the keys (1 in LDS: separate arrays; 4 / 8 in HBM: slots)
};

template <bool G>
__device__ __forceinline__ int32_t ap_cas(int32_t* p, int32_t cmp, int32_t val) {
  return atomicCAS(p, cmp, val);  // ds_cmpst_rtn_b32 on LDS, global_atomic_cmpswap on HBM
}

// In the HBM tier the keys and residues are updated by atomics, which execute in L2: every read of
// them must bypass this CU's L1 (agent-scope relaxed load / exchange), or it may see a stale line.
template <bool G>
__device__ __forceinline__ int32_t ap_key(const ApTable<G>& T, uint32_t s) {
  return G ? __hip_atomic_load(&T.keys[(size_t)(s) * T.ks], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : T.keys[(size_t)(s) * T.ks];
}
template <bool G>
__device__ __forceinline__ double ap_take_residue(const ApTable<G>& T, uint32_t s) {
  if (G) return __hip_atomic_exchange(&T.res[(size_t)(s) * T.ds], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const double r = T.res[(size_t)(s) * T.ds];
  T.res[(size_t)(s) * T.ds] = 0.0;
  return r;
}

// returns the slot of node u, inserting it if absent; 0xFFFFFFFF when the table is full.  The insertion that takes
// the table past `limit` nodes raises *overflow at once: the search is handed to the next tier anyway, and the
// threads that see the flag stop inserting, so the table never fills up - in a full table every further lookup of a
// new node walks all of it (measured in the LDS tier before this check: 4.3 M such walks of 2048 probes each, nine
// tenths of all the probes of the tier)
template <bool G>
__device__ __forceinline__ uint32_t ap_slot(const ApTable<G>& T, int32_t u, uint32_t* used_count, uint32_t limit,
                                            uint32_t* overflow) {
  const uint32_t mask = T.cap - 1;
  uint32_t s = ((uint32_t)u * 2654435761u) >> 7 & mask;
  for (uint32_t probes = 0; probes < T.cap; ++probes) {
    const int32_t k = ap_key<G>(T, s);
    if (k == u) return s;
    if (k == -1) {
      const int32_t prev = ap_cas<G>(&T.keys[(size_t)(s) * T.ks], -1, u);
      if (prev == -1) {
        const uint32_t idx = atomicAdd(used_count, 1u);
        if (idx < T.lcap) T.used[idx] = (typename ApTable<G>::Idx)s;
        if (idx >= limit) *overflow = 1;
        return s;
      }
      if (prev == u) return s;
    }
    s = (s + 1) & mask;
  }
  return 0xFFFFFFFFu;
}

template <bool G>
__global__ __launch_bounds__(256) void k_apbs(const int32_t* __restrict__ target_list, uint32_t t_begin,
                                               uint32_t n_targets, unsigned long long* next_target,
                                               const uint32_t* __restrict__ in_rp, const int32_t* __restrict__ in_ci,
                                               const unsigned long long* __restrict__ out_ext,
                                               const int32_t* __restrict__ old2new, const int32_t* __restrict__ new2old,
                                               double alpha, double rmax, int32_t* __restrict__ out_v,
                                               int32_t* __restrict__ out_t, double* __restrict__ out_p,
                                               unsigned long long out_cap, unsigned long long* out_count,
                                               unsigned long long* out_valid, int32_t* __restrict__ overflow_list, unsigned long long* overflow_count,
                                               unsigned long long* stat_pops, unsigned long long* stat_edges,
                                               char* g_tables, uint32_t g_cap) {
  __shared__ int32_t s_keys[G ? 1 : kApLdsCap];
  __shared__ double s_res[G ? 1 : kApLdsCap];
  __shared__ double s_rsv[G ? 1 : kApLdsCap];
  __shared__ double s_pend[G ? 1 : kApLdsCap];
  __shared__ typename ApTable<G>::Idx s_used[G ? 1 : kApLdsCap];
  __shared__ typename ApTable<G>::Idx s_cur[G ? 1 : kApLdsCap];
  __shared__ typename ApTable<G>::Idx s_nxt[G ? 1 : kApLdsCap];
  __shared__ uint32_t f_row[kApFront];
  __shared__ uint32_t f_off[kApFront + 1];
  __shared__ double f_c[kApFront];
  __shared__ uint32_t s_scan[4];
  __shared__ unsigned long long s_scan64[4];
  __shared__ uint32_t s_used_count, s_nnext, s_overflow;
  __shared__ unsigned long long s_t, s_out_base, s_tot;
  const int tid = threadIdx.x;

  ApTable<G> T;
  using Idx = typename ApTable<G>::Idx;
  if (G) {
    // 32-byte slots + three lists of slot indices that only have to hold the nodes a search may reach before it is
    // handed on (ap_list_cap)
    const uint32_t lcap = ap_list_cap(g_cap);
    const size_t per = (size_t)g_cap * 32 + (size_t)lcap * 12;
    char* base = g_tables + (size_t)blockIdx.x * per;
    // 32-byte slots {residue, reserve, pending, key}: at a quarter load a lookup is one or two probes, so what counts
    // is that the probe, the residue update and the pop of a node touch one 128-byte line, not three
    T.res = (double*)base;
    T.rsv = T.res + 1;
    T.pend = T.res + 2;
    T.keys = (int32_t*)(T.res + 3);
    T.ds = 4;
    T.ks = 8;
    T.used = (Idx*)(base + (size_t)g_cap * 32);
    T.cur = T.used + lcap;
    T.nxt = T.cur + lcap;
    T.cap = g_cap;
    T.lcap = lcap;
  } else {
    T.keys = s_keys; T.res = s_res; T.rsv = s_rsv; T.pend = s_pend;
    T.used = s_used; T.cur = s_cur; T.nxt = s_nxt;
    T.cap = kApLdsCap;
    T.lcap = kApLdsCap;
    T.ds = 1;
    T.ks = 1;
  }
  // LDS tier: give up at 75 % load.  HBM tier: at 25 % - a probe chain is a chain of L2 round trips there, and with
  // linear probing (which keeps a chain inside the 128-byte line its first probe fetched) it is the load factor that
  // decides their length: in a 65 536-slot table at 75 % the counters showed 15 probes per edge on average and
  // 68 for the slowest lane of a wave
  const uint32_t limit = G ? T.cap / kApGlobalLoadDiv : T.cap - T.cap / 4;
  for (uint32_t i = tid; i < T.cap; i += 256) {
    T.keys[(size_t)(i) * T.ks] = -1;
    T.res[(size_t)(i) * T.ds] = 0.0;
    T.rsv[(size_t)(i) * T.ds] = 0.0;
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
        const uint32_t s = ap_slot<G>(T, t, &s_used_count, limit, &s_overflow);
        T.rsv[(size_t)(s) * T.ds] = 1.0;
      }
    } else {
      if (tid == 0) {
        const uint32_t s = ap_slot<G>(T, t, &s_used_count, limit, &s_overflow);
        T.res[(size_t)(s) * T.ds] = 1.0;  // :54-56; the target is pushed unconditionally first
        T.cur[0] = (Idx)s;
      }
      nf = 1;
    }
    __syncthreads();

    while (nf > 0 && !s_overflow) {
      // ---- every frontier node gives up its residue (:58-67,72)
      for (uint32_t i = tid; i < nf; i += 256) {
        const uint32_t s = T.cur[i];
        const double rc = ap_take_residue<G>(T, s);
        T.rsv[(size_t)(s) * T.ds] = T.rsv[(size_t)(s) * T.ds] + rc * alpha;
        T.pend[(size_t)(s) * T.ds] = (1.0 - alpha) * rc;
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
            const int32_t v = ap_key<G>(T, s);
            const uint32_t b = in_rp[v];
            d0 = in_rp[v + 1] - b;
            f_row[i0] = b;
            f_c[i0] = T.pend[(size_t)(s) * T.ds];
          }
          if (i1 < cnt) {
            const uint32_t s = T.cur[fb + i1];
            const int32_t v = ap_key<G>(T, s);
            const uint32_t b = in_rp[v];
            d1 = in_rp[v + 1] - b;
            f_row[i1] = b;
            f_c[i1] = T.pend[(size_t)(s) * T.ds];
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
        // four edges per thread in flight: col_idx loads, then degree gathers, then table updates
        for (uint32_t base = tid; base < E; base += 1024) {
          int32_t u[4];
          double cc[4];
          unsigned long long ext[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const uint32_t e = base + 256u * q;
            u[q] = -1;
            cc[q] = 0.0;
            if (e < E) {
              uint32_t lo = 0, hi = cnt;  // last frontier entry whose edge range starts at or before e
              while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (f_off[mid] <= e) lo = mid + 1; else hi = mid;
              }
              const uint32_t i = lo - 1;
              u[q] = in_ci[f_row[i] + (e - f_off[i])];
              cc[q] = f_c[i];
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) ext[q] = u[q] >= 0 ? out_ext[u[q]] : (1ull << 32);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            if (u[q] < 0 || s_overflow) continue;
            const double add = cc[q] / (double)(uint32_t)(ext[q] >> 32);  // :84-85
            const uint32_t s = ap_slot<G>(T, u[q], &s_used_count, limit, &s_overflow);
            if (s == 0xFFFFFFFFu) {
              s_overflow = 1;
              continue;
            }
            const double old = atomic_add_ret(&T.res[(size_t)(s) * T.ds], add);
            const double nw = old + add;
            if (!(old > rmax) && nw > rmax) {  // :89 strict, un-normalised; first crossing of the level
              const uint32_t pos = atomicAdd(&s_nnext, 1u);
              if (pos < T.lcap) T.nxt[pos] = (Idx)s;
            }
          }
        }
        __syncthreads();
        if (tid == 0 && s_used_count > limit) s_overflow = 1;
        __syncthreads();
      }
      nf = s_nnext;
      Idx* tmp = T.cur; T.cur = T.nxt; T.nxt = tmp;
      __syncthreads();
    }

    const uint32_t used = s_used_count < T.lcap ? s_used_count : T.lcap;
    const bool ovf = s_overflow != 0;
    // ---- emit entries >= threshold (Base_Whole_Graph.java:80-88)
    bool retry = ovf;
    if (!ovf) {
      unsigned long long run = 0;
      for (uint32_t c0 = 0; c0 < used; c0 += 256) {  // count first, one reservation per target
        const uint32_t i = c0 + tid;
        const bool take = i < used && T.rsv[(size_t)T.used[i] * T.ds] > 0.0 && T.rsv[(size_t)T.used[i] * T.ds] >= rmax;
        run += take ? 1ull : 0ull;
      }
      const unsigned long long total = block_sum_u64(run, s_scan64);  // valid in thread 0
      if (tid == 0) {
        s_tot = total;
        s_out_base = total ? atomic_add_u64(out_count, total) : 0ull;
      }
      __syncthreads();
      const unsigned long long tot = s_tot;
      if (s_out_base + tot > out_cap) {
        retry = true;  // the triple buffer is full: the host drains it and runs this target again
        if (tid == 0) atomicMin(out_valid, s_out_base);
      } else {
        unsigned long long at = s_out_base;
        for (uint32_t c0 = 0; c0 < used; c0 += 256) {
          const uint32_t i = c0 + tid;
          const uint32_t s = i < used ? T.used[i] : 0u;
          const bool take = i < used && T.rsv[(size_t)(s) * T.ds] > 0.0 && T.rsv[(size_t)(s) * T.ds] >= rmax;
          unsigned long long chunk_total = 0;
          const unsigned long long ex =
              block_excl_scan_256<unsigned long long>(take ? 1ull : 0ull, s_scan64, &chunk_total);
          if (take) {
            out_v[at + ex] = new2old[ap_key<G>(T, s)];
            out_t[at + ex] = t_old;
            out_p[at + ex] = T.rsv[(size_t)(s) * T.ds];
          }
          at += chunk_total;
        }
      }
    }
    if (retry && tid == 0) {  // table overflow: +t, triple buffer full: -(t + 1)
      const unsigned long long p = atomic_add_u64(overflow_count, 1ull);
      overflow_list[p] = ovf ? t_old : -(t_old + 1);
    }
    __syncthreads();
    // ---- clear the touched slots (all of them after an overflow)
    if (ovf) {
      for (uint32_t i = tid; i < T.cap; i += 256) {
        T.keys[(size_t)(i) * T.ks] = -1;
        T.res[(size_t)(i) * T.ds] = 0.0;
        T.rsv[(size_t)(i) * T.ds] = 0.0;
      }
    } else {
      for (uint32_t i = tid; i < used; i += 256) {
        const uint32_t s = T.used[i];
        T.keys[(size_t)(s) * T.ks] = -1;
        T.res[(size_t)(s) * T.ds] = 0.0;
        T.rsv[(size_t)(s) * T.ds] = 0.0;
      }
    }
    __syncthreads();
  }
  const unsigned long long ps = block_sum_u64(pops, s_scan64);
  const unsigned long long es = block_sum_u64(edges, s_scan64);
  if (tid == 0) {
    if (ps) atomic_add_u64(stat_pops, ps);
    if (es) atomic_add_u64(stat_edges, es);
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
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs<false>)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_apbs<true>)));
  return PPRHIP_OK;
}

int launch_apbs(pprhip_graph* g, bool global_tier, const int32_t* d_targets, uint32_t t_begin, uint32_t n_targets,
                double alpha, double rmax, ApbsBuffers& b) {
  const uint32_t grid = global_tier ? b.g_blocks : std::min<uint32_t>((uint32_t)g->n_cus * 2u, std::max(1u, n_targets));
  if (global_tier)
    k_apbs<true><<<dim3(std::min<uint32_t>(grid, std::max(1u, n_targets))), dim3(256), 0, g->stream>>>(
        d_targets, t_begin, n_targets, b.next_target, g->in_rp, g->in_ci, g->out_ext, g->old2new, g->new2old, alpha, rmax,
        b.out_v, b.out_t, b.out_p, b.out_cap, b.out_count, b.out_valid, b.overflow, b.overflow_count, b.stat_pops, b.stat_edges,
        b.g_tables, b.g_cap);
  else
    k_apbs<false><<<dim3(grid), dim3(256), 0, g->stream>>>(
        d_targets, t_begin, n_targets, b.next_target, g->in_rp, g->in_ci, g->out_ext, g->old2new, g->new2old, alpha, rmax,
        b.out_v, b.out_t, b.out_p, b.out_cap, b.out_count, b.out_valid, b.overflow, b.overflow_count, b.stat_pops, b.stat_edges,
        nullptr, 0);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
