// kernels_push.hip — frontier push kernels for gfx950 (MI355X).
//
// One *level* pushes every frontier node at once from its residue at level start (the
// frontier-synchronous form of Forward_Push.java:86-139 / Backward_Search.java:58-96).  A level
// runs in one of two shapes:
//
//   sparse  k_sparse_prepare (per frontier node: take the residue, credit the reserve, compute the
//           per-edge contribution) then k_sparse_push (edge-parallel over the frontier's edges:
//           frontier entries staged in LDS, coalesced col_idx reads, one returning fp64 atomic per
//           edge, threshold-crossing detection on (old, old + c), crossings collected in LDS and
//           appended to the next frontier with one packed atomic per 2048-edge tile).  Sparse
//           levels are launched in batches: every level's (nodes, edges) counter lives in a small
//           device-side history, and a level's kernels return at once when the previous level left
//           nothing to do (or left so much that the host should switch to the dense shape), so the
//           host reads counters back once per batch, not once per level.
//           (Round 4 tried the batch as ONE launch - prepare and push of up to eight levels inside a kernel whose
//           workgroups meet at a barrier between the steps, shared state read and written past the L2s - and took it
//           back: with grids of 1 to 128 workgroups it was slower everywhere, 2.11-2.23 ms per top-k query against
//           2.04, 0.39-0.55 ms of sparse levels per headline query against 0.34 (profiles/r04_sparse_levels_study.txt).
//           Launches queued back to back overlap their own overhead, and a gated launch costs 3 us; a barrier of
//           memory-side atomics and levels run by fewer workgroups cost more.)
//   dense   k_dense_edges + k_dense_apply + k_dense_reduce: a pull sweep over the non-empty rows of
//           the in-CSR.  Every wave owns 512 consecutive in-edges (8 per lane: two 16-byte column
//           index loads, 8 contribution gathers in flight), sums them by row with a segmented wave
//           scan and stores one value per row; only rows crossing a chunk boundary use an atomic.
//           A streaming kernel then lands each row sum, tests the threshold and prepares crossing
//           rows for the next level (no atomics on the residue vector; per-workgroup counters go
//           to a partials array that a one-workgroup kernel sums).
//
// HBM-bound integer/fp64 work: no MFMA anywhere.  All arithmetic is IEEE double with
// -ffp-contract=off so each product / quotient rounds exactly as the reference's Java does.
#include <algorithm>
#include <cstdlib>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

constexpr int kPushTile = 2048;  // edges per workgroup iteration of k_sparse_push
constexpr int kStageCap = 512;   // frontier entries staged in LDS at a time
constexpr int kCombSlots = 2048;     // LDS table that sums a tile's contributions per destination (power of two)
constexpr int kCombProbes = 4;       // slots tried before an edge goes to memory on its own
constexpr int kCombMinEdges = 262144;  // levels below this many edges skip the table (no gain measured there)

// A batched sparse level runs iff the level before it produced a non-empty frontier that is still
// worth running sparse (the host took that decision itself for the first level of a batch).
__device__ __forceinline__ bool level_runs(unsigned long long pk, int level, unsigned long long dense_thresh) {
  // The first level of a batch always runs: its list can be empty when it was compacted out of a
  // dense level that prepared dead-end nodes only, and their mass still has to land on the source.
  if (level == 0) return true;
  const unsigned long long nf = pk >> kPackShift, ef = pk & kPackMask;
  if (nf == 0) return false;
  return (nf + ef) < dense_thresh;
}

// State of a dense level (GsState): given by the host, or - for a level launched behind another one without a host
// round trip in between - read from the cell the level before it wrote (kGsNone: that level left nothing to sweep).
__device__ __forceinline__ int dense_state(const int* state_in, int state0) { return state_in ? *state_in : state0; }

// ------------------------------------------------------------------------------------------------
// sparse level, step 1: every frontier node gives up its residue
// ------------------------------------------------------------------------------------------------
// (the body of k_sparse_prepare for workgroup `bid` of `nblk`: the one-workgroup kernel that runs several small levels
// in one launch, k_sparse_levels_wg, calls it with 0 of 1; pk: the level's frontier, entries << 36 | edges)
template <int MODE>
__device__ __forceinline__ void sparse_prepare_body(const int32_t* __restrict__ F, const uint32_t* __restrict__ out_rp,
                                                    double* __restrict__ res, double* __restrict__ reserve,
                                                    double* __restrict__ cF, CView c_dense, DevCounters* ctr, int level,
                                                    int dead_slot, unsigned long long pk0, unsigned long long pk,
                                                    const PushArgs& a, uint32_t bid, uint32_t nblk) {
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  // the first level of a batch gets its frontier from the host as an argument (pk0 != ~0) and clears the counters of
  // the levels behind it: no copy and no fill on the stream for what two words and eight zeros say
  if (level == 0 && pk0 != ~0ull && bid == 0 && threadIdx.x == 0) {
    for (int i = 1; i <= kMaxBatch; ++i) ctr->hist[i] = 0ull;
    ctr->hist[kMaxBatch + 2] = 0ull;  // the seeding pass's list counter: the host has read it before this level
  }
  const uint32_t nf = (uint32_t)(pk >> kPackShift);
  double dead = 0.0;
  unsigned long long ndead = 0;
  for (uint32_t i = bid * blockDim.x + threadIdx.x; i < nf; i += nblk * blockDim.x) {
    const int32_t v = F[i];
    const double rc = res[v];
    res[v] = 0.0;                            // Forward_Push.java:89
    reserve[v] = reserve[v] + rc * a.alpha;  // :91-95
    double c;
    if (MODE == kBackward) {
      c = (1.0 - a.alpha) * rc;  // Backward_Search.java:72 (divided by d_out(u) per edge)
    } else {
      const uint32_t d = out_rp[v + 1] - out_rp[v];
      if (d == 0) {  // Forward_Push.java:101-104: the mass goes back to the source
        c = 0.0;
        dead += rc * (1.0 - a.alpha);
        ndead++;
      } else {
        c = ((1.0 - a.alpha) * rc) / (double)d;  // :117
      }
    }
    if (c_dense.p)
      c_dense.at((uint32_t)v) = c;
    else
      cF[i] = c;
  }
  if (MODE != kBackward) {
    const double ds = block_sum_f64(dead, s_red);
    const unsigned long long nd = block_sum_u64(ndead, s_red2);
    if (threadIdx.x == 0 && nd) {
      atomic_add_noret(&ctr->dead[dead_slot], ds);
      atomic_add_u64(&ctr->dead_pops, nd);
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_prepare(const int32_t* __restrict__ F,
                                                         const uint32_t* __restrict__ out_rp,
                                                         double* __restrict__ res, double* __restrict__ reserve,
                                                         double* __restrict__ cF, CView c_dense,
                                                         DevCounters* ctr, int level, unsigned long long dense_thresh,
                                                         int dead_slot, unsigned long long pk0, PushArgs a) {
  const unsigned long long pk = (level == 0 && pk0 != ~0ull) ? pk0 : ctr->hist[level];
  if (!level_runs(pk, level, dense_thresh)) {
    // (the counters behind a first level that does not run - it always runs - need no clearing)
    return;
  }
  sparse_prepare_body<MODE>(F, out_rp, res, reserve, cF, c_dense, ctr, level, dead_slot, pk0, pk, a, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------
// sparse level, step 2: contributions land edge by edge
// ------------------------------------------------------------------------------------------------
struct NewList {  // crossings of the current tile, collected in LDS
  int32_t node[kPushTile + 1];
  uint32_t deg[kPushTile + 1];
  uint32_t count;
};

// One edge lands in three steps so that a thread can keep several edges in flight: the degree
// gather and the returning atomic are issued for a batch of edges before any result is used.
// A node that met the round's threshold at round start without being in the queue ("armed", see engine.hpp) is
// enqueued by the first push that reaches it (Forward_Push.java:226-231 tests the new residue only): whoever clears
// its bit appends it.
__device__ __forceinline__ bool take_armed(uint32_t* __restrict__ armed, int32_t u) {
  const uint32_t bit = 1u << ((uint32_t)u & 31u);
  if (!(armed[(uint32_t)u >> 5] & bit)) return false;
  return (atomicAnd(&armed[(uint32_t)u >> 5], ~bit) & bit) != 0;
}

template <int MODE>
__device__ __forceinline__ void push_finish(int32_t u, double add, double old, uint32_t du,
                                            const uint32_t* __restrict__ in_rp, uint8_t* __restrict__ flags,
                                            uint32_t* __restrict__ armed, NewList* nl, const PushArgs& a) {
  const double nw = old + add;
  bool crossing;
  uint32_t adeg;
  if (MODE == kBackward) {
    crossing = !(old > a.rmax) && (nw > a.rmax);  // Backward_Search.java:89 strict, un-normalised
    adeg = crossing ? in_rp[u + 1] - in_rp[u] : 0u;
  } else {
    const bool was = active_fwd(old, du, a.rmax);
    crossing = !was && active_fwd(nw, du, a.rmax);  // Forward_Push.java:132
    adeg = du;
    if (MODE == kFwdTopk) {
      if (was && a.rmax < a.min_rmax) crossing = take_armed(armed, u);
      if (active_fwd(nw, du, a.min_rmax)) flags[u] = 1;  // :232-237 (parked)
    }
  }
  if (crossing) {
    const uint32_t slot = atomicAdd(&nl->count, 1u);
    nl->node[slot] = u;
    nl->deg[slot] = adeg;
  }
}

template <int MODE>
__device__ __forceinline__ void push_one(int32_t u, double c, const unsigned long long* __restrict__ out_ext,
                                         const uint32_t* __restrict__ in_rp, double* __restrict__ res,
                                         uint8_t* __restrict__ flags, uint32_t* __restrict__ armed, NewList* nl,
                                         const PushArgs& a) {
  const uint32_t du = (uint32_t)(out_ext[u] >> 32);  // packed row extent: one gather for the degree
  const double add = (MODE == kBackward) ? c / (double)du : c;  // Backward_Search.java:84-85
  const double old = atomic_add_ret(&res[u], add);               // Forward_Push.java:123-127
  push_finish<MODE>(u, add, old, du, in_rp, flags, armed, nl, a);
}

// Adds one edge's contribution to its destination's slot of the tile's LDS table (open addressing, a few probes);
// false when no slot could be had, and the edge then lands on its own.
__device__ __forceinline__ bool comb_insert(int32_t* s_key, double* s_val, int32_t u, double add) {
  uint32_t h = ((uint32_t)u * 2654435761u) >> (32 - 11);
  static_assert(kCombSlots == (1 << 11), "hash width follows the table size");
#pragma unroll
  for (int p = 0; p < kCombProbes; ++p) {
    int32_t k = s_key[h];
    if (k == -1) k = atomicCAS(&s_key[h], -1, u);
    if (k == -1 || k == u) {
      __hip_atomic_fetch_add(&s_val[h], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      return true;
    }
    h = (h + 1) & (kCombSlots - 1);
  }
  return false;
}

// Appends the tile's crossings to the next frontier: one packed atomic reserves list slots and
// the edge range, a workgroup scan turns the degrees into edge offsets.
__device__ __forceinline__ void flush_new(NewList* nl, int32_t* __restrict__ Fn, uint32_t* __restrict__ eoffn,
                                          unsigned long long* out_counter) {
  __shared__ unsigned long long s_scan[4];
  __shared__ unsigned long long s_base;
  __syncthreads();
  const uint32_t cnt = nl->count;
  if (cnt == 0) return;  // uniform
  const int tid = threadIdx.x;
  // each thread owns up to 9 consecutive collected entries (2049 / 256 rounded up)
  constexpr int kPer = (kPushTile + 1 + 255) / 256;
  const uint32_t b = tid * kPer;
  unsigned long long mine = 0;
#pragma unroll
  for (int j = 0; j < kPer; ++j)
    if (b + j < cnt) mine += nl->deg[b + j];
  unsigned long long total = 0;
  const unsigned long long excl = block_excl_scan_256<unsigned long long>(mine, s_scan, &total);
  if (tid == 0) s_base = atomic_add_u64(out_counter, ((unsigned long long)cnt << kPackShift) | total);
  __syncthreads();
  const uint32_t pos0 = (uint32_t)(s_base >> kPackShift);
  unsigned long long e = (s_base & kPackMask) + excl;
#pragma unroll
  for (int j = 0; j < kPer; ++j)
    if (b + j < cnt) {
      Fn[pos0 + b + j] = nl->node[b + j];
      eoffn[pos0 + b + j] = (uint32_t)e;
      e += nl->deg[b + j];
    }
  __syncthreads();
  if (tid == 0) nl->count = 0;
  __syncthreads();
}

// (the body of k_sparse_push for workgroup `bid` of `nblk`, see sparse_prepare_body)
template <int MODE>
__device__ __forceinline__ void sparse_push_body(const int32_t* __restrict__ F, const double* __restrict__ cF,
                                                 const uint32_t* __restrict__ eoff, const uint32_t* __restrict__ trp,
                                                 const int32_t* __restrict__ tci,
                                                 const unsigned long long* __restrict__ out_ext,
                                                 const uint32_t* __restrict__ in_rp, double* __restrict__ res,
                                                 uint8_t* __restrict__ flags, uint32_t* __restrict__ armed,
                                                 int32_t* __restrict__ Fn, uint32_t* __restrict__ eoffn, DevCounters* ctr,
                                                 int level, int dead_slot, unsigned long long comb_min,
                                                 unsigned long long pk, const PushArgs& a, uint32_t bid, uint32_t nblk) {
  __shared__ uint32_t s_eoff[kStageCap + 1];
  __shared__ uint32_t s_row[kStageCap];
  __shared__ double s_c[kStageCap];
  __shared__ uint32_t s_i0;
  __shared__ NewList s_new;
  __shared__ int32_t s_key[kCombSlots];
  __shared__ double s_val[kCombSlots];
  const int tid = threadIdx.x;
  const uint32_t nf = (uint32_t)(pk >> kPackShift);
  const unsigned long long E = pk & kPackMask;
  unsigned long long* out_counter = &ctr->hist[level + 1];
  // Levels with enough edges to repeat destinations (hubs collect a large share of any R-MAT frontier's edges) sum
  // a tile's contributions per destination in LDS first, so a destination costs one returning global atomic and one
  // degree gather per tile instead of one per edge; small levels go straight to memory.
  const bool combine = E >= comb_min;
  if (tid == 0) s_new.count = 0;
  if (combine)
    for (int j = tid; j < kCombSlots; j += 256) {
      s_key[j] = -1;
      s_val[j] = 0.0;
    }
  __syncthreads();

  if (MODE != kBackward && bid == 0) {
    // dead-end mass of this level lands on the source (Forward_Push.java:101-113)
    if (tid == 0) {
      const double dead = ctr->dead[dead_slot];
      if (dead > 0.0) {
        push_one<MODE>(a.src, dead, out_ext, in_rp, res, flags, armed, &s_new, a);
        ctr->dead[dead_slot] = 0.0;
      }
    }
    flush_new(&s_new, Fn, eoffn, out_counter);
  }

  const unsigned long long n_tiles = (E + kPushTile - 1) / kPushTile;
  for (unsigned long long t = bid; t < n_tiles; t += nblk) {
    const unsigned long long tile_lo = t * kPushTile;
    const unsigned long long tile_hi = (tile_lo + kPushTile < E) ? tile_lo + kPushTile : E;
    if (tid == 0) {  // last frontier index whose edge range starts at or before tile_lo
      uint32_t lo = 0, hi = nf;
      while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((unsigned long long)eoff[mid] <= tile_lo) lo = mid + 1; else hi = mid;
      }
      s_i0 = lo - 1;
    }
    __syncthreads();
    uint32_t ci0 = s_i0;
    unsigned long long ce = tile_lo;
    while (ce < tile_hi) {
      const uint32_t cnt = (nf - ci0 < (uint32_t)kStageCap) ? nf - ci0 : (uint32_t)kStageCap;
      if (cnt == 0) break;
      for (uint32_t j = tid; j <= cnt; j += 256) {
        const uint32_t idx = ci0 + j;
        s_eoff[j] = idx < nf ? eoff[idx] : (uint32_t)E;
        if (j < cnt) {
          s_row[j] = trp[F[idx]];
          s_c[j] = cF[idx];
        }
      }
      __syncthreads();
      const unsigned long long cov_hi = ((unsigned long long)s_eoff[cnt] < tile_hi) ? s_eoff[cnt] : tile_hi;
      // four edges per thread in flight: col_idx loads, then degree gathers, then atomics, then tests
      for (unsigned long long base = ce + tid; base < cov_hi; base += 1024) {
        int32_t u[4];
        double add[4], old[4];
        uint32_t du[4];
        bool valid[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned long long e = base + 256ull * q;
          valid[q] = e < cov_hi;
          u[q] = 0;
          add[q] = 0.0;
          if (valid[q]) {
            const uint32_t e32 = (uint32_t)e;
            uint32_t lo = 0, hi = cnt;  // last staged entry whose range starts at or before e
            while (lo < hi) {
              const uint32_t mid = (lo + hi) >> 1;
              if (s_eoff[mid] <= e32) lo = mid + 1; else hi = mid;
            }
            const uint32_t j = lo - 1;
            u[q] = tci[s_row[j] + (e32 - s_eoff[j])];
            add[q] = s_c[j];
          }
        }
        if (MODE == kBackward || !combine) {
#pragma unroll
          for (int q = 0; q < 4; ++q) du[q] = valid[q] ? (uint32_t)(out_ext[u[q]] >> 32) : 1u;
        }
        if (MODE == kBackward) {  // every edge's quotient rounds on its own (Backward_Search.java:84-85)
#pragma unroll
          for (int q = 0; q < 4; ++q) add[q] = add[q] / (double)du[q];
        }
        if (combine) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (valid[q] && comb_insert(s_key, s_val, u[q], add[q])) valid[q] = false;  // lands with its slot
          if (MODE != kBackward) {
#pragma unroll
            for (int q = 0; q < 4; ++q) du[q] = valid[q] ? (uint32_t)(out_ext[u[q]] >> 32) : 1u;  // table full
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          old[q] = 0.0;
          if (valid[q]) old[q] = atomic_add_ret(&res[u[q]], add[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (valid[q]) push_finish<MODE>(u[q], add[q], old[q], du[q], in_rp, flags, armed, &s_new, a);
      }
      __syncthreads();
      ce = cov_hi;
      ci0 += cnt;
    }
    if (combine) {
      // the tile's per-destination sums land: four slots per thread in flight, slots left empty for the next tile
      for (int j0 = tid; j0 < kCombSlots; j0 += 1024) {
        int32_t u[4];
        double add[4], old[4];
        uint32_t du[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          u[q] = s_key[j0 + 256 * q];
          add[q] = s_val[j0 + 256 * q];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) du[q] = u[q] >= 0 ? (uint32_t)(out_ext[u[q]] >> 32) : 1u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          old[q] = 0.0;
          if (u[q] >= 0) {
            old[q] = atomic_add_ret(&res[u[q]], add[q]);
            s_key[j0 + 256 * q] = -1;
            s_val[j0 + 256 * q] = 0.0;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (u[q] >= 0) push_finish<MODE>(u[q], add[q], old[q], du[q], in_rp, flags, armed, &s_new, a);
      }
    }
    flush_new(&s_new, Fn, eoffn, out_counter);
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_push(const int32_t* __restrict__ F, const double* __restrict__ cF,
                                                      const uint32_t* __restrict__ eoff,
                                                      const uint32_t* __restrict__ trp, const int32_t* __restrict__ tci,
                                                      const unsigned long long* __restrict__ out_ext,
                                                      const uint32_t* __restrict__ in_rp, double* __restrict__ res,
                                                      uint8_t* __restrict__ flags, uint32_t* __restrict__ armed,
                                                      int32_t* __restrict__ Fn,
                                                      uint32_t* __restrict__ eoffn, DevCounters* ctr, int level,
                                                      unsigned long long dense_thresh, int dead_slot,
                                                      unsigned long long comb_min, unsigned long long pk0, PushArgs a) {
  const unsigned long long pk = (level == 0 && pk0 != ~0ull) ? pk0 : ctr->hist[level];
  if (!level_runs(pk, level, dense_thresh)) return;
  sparse_push_body<MODE>(F, cF, eoff, trp, tci, out_ext, in_rp, res, flags, armed, Fn, eoffn, ctr, level, dead_slot, comb_min,
                         pk, a, blockIdx.x, gridDim.x);
}

// Several SMALL sparse levels in one launch, on one workgroup: levels first .. last of a batch, each as long as the
// level before it left a frontier that is worth a sparse level (level_runs) and small enough for one workgroup
// (entries + edges < wg_cap; the first level is the host's choice).  A top-k round's push is 8.4 levels on R-MAT 22, three
// quarters of them below 4 096 entries + edges (60 % below 512), and each cost two launches and ~21 us of stream time
// whatever its size; here a level costs its dependent memory accesses.  Between the levels (and between a level's two
// steps) the workgroup's waves meet at a barrier with a release fence before it and an acquire fence behind it: what one
// wave wrote - list entries, contributions, counters - the others read from L2.  The host reads the batch's counters
// as before and applies the same two rules to tell which levels ran.
template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_levels_wg(int32_t* F0, int32_t* F1, uint32_t* eoff0, uint32_t* eoff1,
                                                           const uint32_t* __restrict__ out_rp, double* res,
                                                           double* reserve, double* cF, const uint32_t* __restrict__ trp,
                                                           const int32_t* __restrict__ tci,
                                                           const unsigned long long* __restrict__ out_ext,
                                                           const uint32_t* __restrict__ in_rp, uint8_t* flags,
                                                           uint32_t* armed, DevCounters* ctr, int fb0, int first, int last,
                                                           unsigned long long dense_thresh, unsigned long long wg_cap,
                                                           int dead_slot, unsigned long long comb_min,
                                                           unsigned long long pk0, PushArgs a) {
  for (int level = first; level <= last; ++level) {
    const unsigned long long pk =
        (level == 0 && pk0 != ~0ull)
            ? pk0
            : __hip_atomic_load(&ctr->hist[level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!level_runs(pk, level, dense_thresh)) return;
    if (level > 0 && (pk >> kPackShift) + (pk & kPackMask) >= wg_cap) return;  // too large for one workgroup
    const int fb = fb0 ^ (level & 1);
    int32_t* const F = fb ? F1 : F0;
    int32_t* const Fn = fb ? F0 : F1;
    uint32_t* const eo = fb ? eoff1 : eoff0;
    uint32_t* const eon = fb ? eoff0 : eoff1;
    sparse_prepare_body<MODE>(F, out_rp, res, reserve, cF, CView{nullptr, 1, 0}, ctr, level, dead_slot, pk0, pk, a, 0u, 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    sparse_push_body<MODE>(F, cF, eo, trp, tci, out_ext, in_rp, res, flags, armed, Fn, eon, ctr, level, dead_slot, comb_min, pk,
                           a, 0u, 1u);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
}

// ------------------------------------------------------------------------------------------------
// dense level: pull sweep over the non-empty rows of the in-CSR
// ------------------------------------------------------------------------------------------------
// k_dense_edges: one wave per chunk of 512 consecutive in-edges, 8 per lane.  A lane reads its 8
// column indices as two 16-byte loads and one byte of row-start flags, gathers the 8 contributions
// and sums them by row; rows that end inside the wave are completed with a segmented wave scan and
// stored, only the (at most two) rows that cross the chunk boundary use an fp64 atomic.  No LDS,
// no workgroup barrier, no special case for hub rows: every wave carries the same 512 gathers.
constexpr int kChunkEdges = 512;
constexpr int kHotMax = 16384;  // contributions of the 16K highest-out-degree vertices live in LDS (128 KB)

// The internal vertex order puts the highest out-degrees first (graph lift), so ids < n_hot are the
// contributions gathered most often (42 % of all in-edges at R-MAT scale 22).  A persistent workgroup
// per CU stages them in LDS once per level and serves those gathers from LDS; from L2 every 8-byte
// value costs the L1 a 128-byte line fill, and that line path is what bounds this kernel otherwise
// (250 G gathers/s when everything hits L2; the LDS table buys 19 %).
struct ChunkRegs {  // one lane's share of a chunk: 8 column indices + their row-start flags
  int4 ia, ib;
  uint32_t fb;
};

__device__ __forceinline__ ChunkRegs load_chunk(const int32_t* __restrict__ in_ci,
                                                const uint8_t* __restrict__ start_flags, uint32_t c, int lane) {
  const unsigned long long e0 = (unsigned long long)c * kChunkEdges + 8ull * lane;
  // read once per sweep: non-temporal, so that the index stream does not push gathered lines out of L2
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i* q = reinterpret_cast<const v4i*>(in_ci + e0);
  const v4i x = __builtin_nontemporal_load(q), y = __builtin_nontemporal_load(q + 1);
  ChunkRegs r;
  r.ia = make_int4(x.x, x.y, x.z, x.w);
  r.ib = make_int4(y.x, y.y, y.z, y.w);
  r.fb = __builtin_nontemporal_load(&start_flags[e0 >> 3]);
  return r;
}

// Window of a chunk in the launch's virtual order (engine.hpp: EdgeWindows); w only moves forward.
__device__ __forceinline__ uint32_t window_chunk(const EdgeWindows& W, uint32_t vc, uint32_t* w) {
  uint32_t x = *w;
  while (x + 1 < W.n && vc >= W.c_pre[x + 1]) ++x;
  *w = x;
  return W.c_lo[x] + (vc - W.c_pre[x]);
}

// SLICED: the edge arrays are the sliced copy (engine.hpp: SlicedLayout): a flag starts a *segment*, seg_row maps it
// to its row ordinal, and every segment sum is added to the row's accumulator with an fp64 atomic (a row has one
// segment per slice; k_dense_apply leaves the accumulators zero).  Otherwise segments are rows and a row that starts
// and ends inside a chunk is stored.
template <bool HOT, bool SLICED>
__global__ __launch_bounds__(1024) void k_dense_edges(const int32_t* __restrict__ in_ci,
                                                       const uint8_t* __restrict__ start_flags,
                                                       const uint32_t* __restrict__ chunk_starts,
                                                       const uint32_t* __restrict__ seg_row, EdgeWindows W,
                                                       const double* __restrict__ c_cur,
                                                       double* __restrict__ acc_nz, uint32_t n_hot,
                                                       const int* state_in) {
  // One block of a sweep: the chunks that hold the in-edges of the block's rows (the whole CSR when the sweep is not
  // cut into blocks), as a list of windows.  Edges of a boundary chunk outside the window count as zero: the launch
  // (or window) they belong to sums them.
  extern __shared__ __attribute__((aligned(16))) double s_hot[];
  if (dense_state(state_in, kGsJacobi) == kGsNone) return;
  const int lane = lane_id();
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t stride = gridDim.x * waves_per_block;
  const uint32_t total = W.c_pre[W.n];
  uint32_t vc = blockIdx.x * waves_per_block + (uint32_t)__builtin_amdgcn_readfirstlane(wave_id());
  uint32_t w = 0, c = 0;
  ChunkRegs cur;
  if (vc < total) {
    c = window_chunk(W, vc, &w);
    cur = load_chunk(in_ci, start_flags, c, lane);  // in flight while the hot table loads
  }
  if (HOT) {
    // 16 values per thread, loaded in one batch
    double t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t i = threadIdx.x + j * 1024u;
      t[j] = i < n_hot ? c_cur[i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t i = threadIdx.x + j * 1024u;
      if (i < n_hot) s_hot[i] = t[j];
    }
    __syncthreads();
  }
  for (; vc < total; vc += stride) {
    // next chunk's indices are requested before this chunk's gathers, so their latency is hidden
    ChunkRegs nxt = cur;
    uint32_t wn = w, cn = c;
    if (vc + stride < total) {
      cn = window_chunk(W, vc + stride, &wn);
      nxt = load_chunk(in_ci, start_flags, cn, lane);
    }
    const unsigned long long e_lo = W.e_lo[w], e_hi = W.e_hi[w];
    const uint32_t cs = chunk_starts[c];
    const unsigned long long e0 = (unsigned long long)c * kChunkEdges + 8ull * lane;
    const uint32_t fb = cur.fb;
    const int32_t idx[8] = {cur.ia.x, cur.ia.y, cur.ia.z, cur.ia.w, cur.ib.x, cur.ib.y, cur.ib.z, cur.ib.w};
    double v[8];
    if (HOT) {
      // branch-free: every lane issues both loads (hot lanes read c_cur[0], one shared line; cold
      // lanes read s_hot[0]) so that all 8 global gathers of the lane stay in flight together
      double gl[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) gl[i] = c_cur[(uint32_t)idx[i] < n_hot ? 0 : idx[i]];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const double hv = s_hot[(uint32_t)idx[i] < n_hot ? idx[i] : 0];
        v[i] = (uint32_t)idx[i] < n_hot ? hv : gl[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = c_cur[idx[i]];
    }
    if (e0 < e_lo || e0 + 8 > e_hi) {  // first / last chunk of the block only
#pragma unroll
      for (int i = 0; i < 8; ++i)
        if (e0 + i < e_lo || e0 + i >= e_hi) v[i] = 0.0;
    }
    // row index of a segment = (row starts at or before its first edge) - 1
    const uint32_t pc = __popc(fb);
    const uint32_t incl = wave_incl_scan_u32_dpp(pc);
    const uint32_t before = cs + incl - pc;  // row starts before this lane's first edge
    double seg = 0.0, first_seg = 0.0;
    uint32_t k = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((fb >> i) & 1u) {
        if (k == 0) {
          first_seg = seg;  // closes the row carried in from earlier lanes
        } else if (SLICED) {
          if (seg != 0.0) atomic_add_noret(&acc_nz[seg_row[before + k - 1]], seg);
        } else {
          acc_nz[before + k - 1] = seg;  // a row that starts and ends inside this lane
        }
        seg = 0.0;
        ++k;
      }
      seg += v[i];
    }
    // segmented scan over lanes: S(l) = x(l) + (lane l holds a row start ? 0 : S(l-1))
    const bool h = k != 0;
    const double sval = wave_seg_scan_f64_dpp(seg, h);
    const double carry = wave_prev_f64_dpp(sval);
    const unsigned long long hmask = __ballot(h);
    if (h) {
      // the row that ends at this lane's first start flag: edges carried in + this lane's head
      const bool nonempty = lane > 0 || (fb & 1u) == 0;
      if (nonempty && before > 0) {
        const double tot = carry + first_seg;
        const bool started_here = (hmask & ((1ull << lane) - 1ull)) != 0;  // an earlier lane starts a row
        if (SLICED) {
          if (tot != 0.0) atomic_add_noret(&acc_nz[seg_row[before - 1]], tot);
        } else if (started_here) {
          acc_nz[before - 1] = tot;
        } else {
          atomic_add_noret(&acc_nz[before - 1], tot);  // began in an earlier chunk
        }
      }
    }
    if (lane == 63) {  // the row still open at the end of the chunk
      const uint32_t starts = cs + incl;
      if (starts > 0 && sval != 0.0) atomic_add_noret(&acc_nz[SLICED ? seg_row[starts - 1] : starts - 1], sval);
    }
    cur = nxt;
    w = wn;
    c = cn;
  }
}

// ------------------------------------------------------------------------------------------------
// batched dense level: kBatch queries per sweep
// ------------------------------------------------------------------------------------------------
// A gather moves a 128-byte line whether 8 bytes of it are used or all of it, and the rate of lines
// that leave L2 (52-55 G/s = the HBM roof at line granularity) is what bounds the sweep: the edge
// kernel below runs at 51.7 G lines/s.  The batched sweep keeps the contributions of
// kBatch = 16 concurrent queries interleaved, c8[v][slot] (128 bytes per vertex = one L2 line), so
// the line a gather brings in carries that vertex's contribution for every query in flight, and
// the column indices are read once for all of them.  G = kBatch lanes (one per slot) share an
// edge: a wave still owns a 512-edge chunk, lane group g = lane / G walks edges [8Gg, 8G(g+1)) of
// it in order.  The group's column indices sit in its own lanes' registers (two coalesced 16-byte
// loads per lane) and are broadcast inside the group with ds_swizzle; row sums close inside the
// group where a row starts and ends there, cross groups with a short segmented scan, and only rows
// crossing the chunk boundary use atomics.  (Measured on R-MAT 22, all slots busy: 0.69 ms per
// sweep at G = 8, 0.83 ms at G = 16, 1.85 ms at G = 32.)
constexpr int kHotBytes = 128 * 1024;  // LDS table of the hottest vertices' lines (2048 x 64 B or 1024 x 128 B)
// The batched edge kernel takes 32 KB of it (256 lines) since round 5: the table's size never mattered to the sweep
// itself (0 / 256 / 512 / 1024 lines within 0.5 %, round 2), but a workgroup that holds 128 of a CU's 160 KB keeps
// every kernel with a larger LDS block of its own - the sparse push's 48 KB - off the CU while it runs, and the
// queries that work beside the sweeps (fora.cpp: SlotDriver) wait for the gaps between the sweep's kernels:
// k_sparse_push took 99 us per launch beside the sweeps against 14 us alone.  128 -> 32 KB: 344-347 -> 352 queries/s.
constexpr int kHotDefaultBytes = 32 * 1024;

// value of lane K of the caller's lane group (G = 8 or 16 lanes)
template <int G, int K>
__device__ __forceinline__ int group_bcast(int x) {
  return __builtin_amdgcn_ds_swizzle(x, (0x1f & ~(G - 1)) | (K << 5));
}

template <int G>
struct ChunkRegsB {
  int4 ia, ib;
  unsigned long long mask[G / 8];  // row-start bits of the lane group's 8 * G edges
};

template <int G>
__device__ __forceinline__ ChunkRegsB<G> load_chunk_b(const int32_t* __restrict__ in_ci,
                                                      const unsigned long long* __restrict__ flags64, uint32_t c,
                                                      int lane) {
  const unsigned long long e0 = (unsigned long long)c * kChunkEdges + 8ull * lane;
  const int4* p = reinterpret_cast<const int4*>(in_ci + e0);
  ChunkRegsB<G> r;
  // the index stream is read once per sweep: non-temporal, so that it does not push gathered lines out of L2
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i* q = reinterpret_cast<const v4i*>(p);
  const v4i x = __builtin_nontemporal_load(q), y = __builtin_nontemporal_load(q + 1);
  r.ia = make_int4(x.x, x.y, x.z, x.w);
  r.ib = make_int4(y.x, y.y, y.z, y.w);
#pragma unroll
  for (int w = 0; w < G / 8; ++w)
    r.mask[w] = __builtin_nontemporal_load(&flags64[(size_t)c * 8 + (size_t)(lane / G) * (G / 8) + w]);
  return r;
}

// 8 edges of the group: the indices sit in lane JB of the group.
template <bool HOT, int G, int JB>
__device__ __forceinline__ void edges_b_block(const ChunkRegsB<G>& cur, const double* __restrict__ cB,
                                              const double* s_hot, uint32_t n_hot, int s, bool tail,
                                              unsigned long long e_first, unsigned long long e_lo,
                                              unsigned long long e_hi, uint32_t before,
                                              double* __restrict__ accB, double& seg, double& first_seg, uint32_t& k) {
  const int32_t own[8] = {cur.ia.x, cur.ia.y, cur.ia.z, cur.ia.w, cur.ib.x, cur.ib.y, cur.ib.z, cur.ib.w};
  uint32_t v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (uint32_t)group_bcast<G, JB>(own[i]);
  double val[8];
  if (HOT) {
    double gl[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) gl[i] = cB[(size_t)(v[i] < n_hot ? 0u : v[i]) * G + s];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const double hv = s_hot[(v[i] < n_hot ? v[i] : 0u) * G + s];
      val[i] = v[i] < n_hot ? hv : gl[i];
    }
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) val[i] = cB[(size_t)v[i] * G + s];
  }
  if (tail) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (e_first + JB * 8 + i < e_lo || e_first + JB * 8 + i >= e_hi) val[i] = 0.0;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if ((cur.mask[(JB * 8 + i) >> 6] >> ((JB * 8 + i) & 63)) & 1ull) {
      if (k == 0)
        first_seg = seg;  // closes the row carried in from earlier groups
      else
        __builtin_nontemporal_store(seg, &accB[(size_t)(before + k - 1) * G + s]);  // a row that starts and ends inside this group
      seg = 0.0;
      ++k;
    }
    seg += val[i];
  }
}

template <bool HOT, int G, int JB>
struct EdgeBlocks {
  static __device__ __forceinline__ void run(const ChunkRegsB<G>& cur, const double* __restrict__ cB,
                                             const double* s_hot, uint32_t n_hot, int s, bool tail,
                                             unsigned long long e_first, unsigned long long e_lo,
                                             unsigned long long e_hi, uint32_t before,
                                             double* __restrict__ accB, double& seg, double& first_seg, uint32_t& k) {
    EdgeBlocks<HOT, G, JB - 1>::run(cur, cB, s_hot, n_hot, s, tail, e_first, e_lo, e_hi, before, accB, seg, first_seg, k);
    edges_b_block<HOT, G, JB>(cur, cB, s_hot, n_hot, s, tail, e_first, e_lo, e_hi, before, accB, seg, first_seg, k);
  }
};
template <bool HOT, int G>
struct EdgeBlocks<HOT, G, -1> {
  static __device__ __forceinline__ void run(const ChunkRegsB<G>&, const double*, const double*, uint32_t, int, bool,
                                             unsigned long long, unsigned long long, unsigned long long, uint32_t,
                                             double*, double&, double&, uint32_t&) {}
};

// G = queries per sweep = lanes per edge; the wave's 64 / G lane groups walk 8 * G edges each.
template <bool HOT, int G>
__global__ __launch_bounds__(1024) void k_dense_edges_b(const int32_t* __restrict__ in_ci,
                                                         const unsigned long long* __restrict__ flags64,
                                                         const uint32_t* __restrict__ chunk_starts, uint32_t n_chunks,
                                                         unsigned long long m, const double* __restrict__ cB,
                                                         double* __restrict__ accB, uint32_t n_hot, uint32_t c_lo,
                                                         unsigned long long e_lo, unsigned long long e_hi, uint32_t n) {
  // one block of a sweep: chunks [c_lo, n_chunks) holding the in-edges [e_lo, e_hi) (see k_dense_edges)
  extern __shared__ __attribute__((aligned(16))) double s_hot[];
  const int lane = lane_id();
  const int grp = lane / G, s = lane & (G - 1);
  const uint32_t waves_per_block = blockDim.x >> 6;
  const uint32_t stride = gridDim.x * waves_per_block;
  uint32_t c = c_lo + blockIdx.x * waves_per_block + (uint32_t)wave_id();
  ChunkRegsB<G> cur;
  if (c < n_chunks) cur = load_chunk_b<G>(in_ci, flags64, c, lane);
  if (HOT) {
    double t[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t i = threadIdx.x + j * 1024u;
      t[j] = i < n_hot * G ? cB[i] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const uint32_t i = threadIdx.x + j * 1024u;
      if (i < n_hot * G) s_hot[i] = t[j];
    }
    __syncthreads();
  }
  for (; c < n_chunks; c += stride) {
    ChunkRegsB<G> nxt = cur;
    const uint32_t cn = c + stride;
    if (cn < n_chunks) nxt = load_chunk_b<G>(in_ci, flags64, cn, lane);
    const uint32_t cs = chunk_starts[c];
    uint32_t pc = 0;
#pragma unroll
    for (int w = 0; w < G / 8; ++w) pc += (uint32_t)__popcll(cur.mask[w]);
    const uint32_t incl = wave_incl_scan_u32_dpp(s == 0 ? pc : 0u);  // row starts up to and including this group
    const uint32_t before = cs + incl - pc;
    const unsigned long long e_first = (unsigned long long)c * kChunkEdges + (unsigned long long)(8 * G) * grp;
    const unsigned long long c_e0 = (unsigned long long)c * kChunkEdges;
    const bool tail = c_e0 < e_lo || c_e0 + kChunkEdges > e_hi;  // first / last chunk of the block: edges outside count 0
    double seg = 0.0, first_seg = 0.0;
    uint32_t k = 0;
    EdgeBlocks<HOT, G, G - 1>::run(cur, cB, s_hot, n_hot, s, tail, e_first, e_lo, e_hi, before, accB, seg, first_seg, k);
    // segmented scan over the lane groups: S(g) = tail(g) + (group g holds a row start ? 0 : S(g-1))
    const bool h = k != 0;
    double S = seg;
    int F = h ? 1 : 0;
#pragma unroll
    for (int d = G; d < 64; d <<= 1) {
      const double ps = __shfl_up(S, d);
      const int pf = __shfl_up(F, d);
      if (lane >= d) {
        if (!F) S += ps;
        F |= pf;
      }
    }
    double carry = __shfl_up(S, G);
    if (lane < G) carry = 0.0;
    const unsigned long long hmask = __ballot(h);
    if (h) {
      const bool nonempty = grp > 0 || (cur.mask[0] & 1ull) == 0;
      if (nonempty && before > 0) {
        const double total = carry + first_seg;
        const bool started_here = (hmask & ((1ull << (grp * G)) - 1ull)) != 0;
        double* dst = &accB[(size_t)(before - 1) * G + s];
        if (started_here)
          *dst = total;
        else
          atomic_add_noret(dst, total);  // began in an earlier chunk
      }
    }
    if (grp == 64 / G - 1) {  // the row still open at the end of the chunk
      const uint32_t starts = cs + incl;
      if (starts > 0 && S != 0.0) atomic_add_noret(&accB[(size_t)(starts - 1) * G + s], S);
    }
    cur = nxt;
  }
}

// ------------------------------------------------------------------------------------------------
// k_dense_edges_panel (round 6): the single-query forward edge kernel over the row-panel copy (engine_internal.hpp:
// HostPanelLayout).  A workgroup takes an ITEM - at most kItemEdges edges of one panel of kPanelRows rows, sorted by
// source - and sums it into acc[row] in LDS (128 KB).  A wave takes 512 consecutive edges per turn, lane l the edges l,
// l + 64, ... of them (stored as the lane's 32 bytes), the workgroup 8 192: neighbouring lanes gather neighbouring sources, so the sixteen contributions of a 128-byte line are
// one request to the L1 (which keeps ~256 lines in flight per CU - what bounds the row-major and the sliced kernel:
// TCP_PENDING_STALL_CYCLES 0.69 of their cycles), and every workgroup walks the contribution array front to back.  Sums
// land with ds_add_f64 (zero contributions are skipped: a dense level's frontier is a part of the nodes); at the end
// the item's rows leave as one contiguous block part[part0 + row]; k_dense_apply<.., true> adds a row's parts.
// Rows outside [j_lo, j_hi) - the Gauss-Seidel block of the launch, whose bounds may cut a panel - are left out.
// Items are dealt to the workgroups in turn (they are of one size: the parts of a panel hold equal edge counts).
// ------------------------------------------------------------------------------------------------
constexpr int kPanelThreads = 1024;
constexpr int kPanelLdsBytes = (int)(kPanelRows * sizeof(double));
static_assert(kPanelStep == (uint32_t)kPanelThreads * 8u, "eight edges per lane and turn");

__global__ __launch_bounds__(kPanelThreads, 8) void k_dense_edges_panel(const int32_t* __restrict__ src,
                                                                     const uint16_t* __restrict__ rloc,
                                                                     const PanelItem* __restrict__ items,
                                                                     uint32_t item_lo, uint32_t item_hi,
                                                                     const double* __restrict__ c_cur,
                                                                     double* __restrict__ part, uint32_t j_lo,
                                                                     uint32_t j_hi, uint32_t n_nz, const int* state_in,
                                                                     uint32_t* __restrict__ queue) {
  extern __shared__ __attribute__((aligned(16))) double acc[];
  __shared__ uint32_t s_take;
  typedef int v4i __attribute__((ext_vector_type(4)));
  if (dense_state(state_in, kGsJacobi) == kGsNone) return;  // (the queue stays at zero: k_dense_reduce left it so)
  const uint32_t tid = threadIdx.x;
  // items are handed out by a counter (they differ in size by the panels' rounding, and the last round of a static deal
  // would leave most CUs idle); the level's k_dense_reduce zeroes it behind the launch
  uint32_t it = item_lo + blockIdx.x;
  while (it < item_hi) {
    const PanelItem I = items[it];
    if (tid == 0) s_take = atomicAdd(queue, 1u);
    {
      double2* a2 = reinterpret_cast<double2*>(acc);
#pragma unroll
      for (int k = 0; k < (int)(kPanelRows / 2 / kPanelThreads); ++k) a2[(uint32_t)k * kPanelThreads + tid] = make_double2(0.0, 0.0);
    }
    __syncthreads();
    const uint32_t row0 = I.panel * kPanelRows;
    const uint32_t r_lo = j_lo > row0 ? j_lo - row0 : 0u;
    const uint32_t r_hi = j_hi > row0 ? min(min(j_hi, n_nz) - row0, kPanelRows) : 0u;  // (padding: row 0xffff >= r_hi)
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4i* sp = reinterpret_cast<const v4i*>(src + (size_t)I.edge0 * kPanelStep) + 2u * tid;
    const v4u* rp = reinterpret_cast<const v4u*>(rloc + (size_t)I.edge0 * kPanelStep) + tid;
    // the index streams are read once per sweep: non-temporal, so that they do not push gathered lines out of L2
    // (two turns ahead: a turn's index loads are asked for behind the gathers of the turn before the last, so that the
    // wait for a turn's gathers - loads return in order - never waits for the stream from HBM)
    v4i ia = __builtin_nontemporal_load(sp), ib = __builtin_nontemporal_load(sp + 1);
    v4u rx = __builtin_nontemporal_load(rp);
    v4i na = ia, nb = ib;
    v4u nr = rx;
    if (I.steps > 1) {
      na = __builtin_nontemporal_load(sp + (size_t)(2 * kPanelThreads));
      nb = __builtin_nontemporal_load(sp + (size_t)(2 * kPanelThreads) + 1);
      nr = __builtin_nontemporal_load(rp + (size_t)kPanelThreads);
    }
    for (uint32_t i = 0; i < I.steps; ++i) {
      const int32_t u[8] = {ia.x, ia.y, ia.z, ia.w, ib.x, ib.y, ib.z, ib.w};
      const uint32_t r[8] = {rx.x & 0xffffu, rx.x >> 16, rx.y & 0xffffu, rx.y >> 16,
                             rx.z & 0xffffu, rx.z >> 16, rx.w & 0xffffu, rx.w >> 16};
      bool in[8];
      double v[8];
      // (rows outside the block and the padding gather the first contribution - one shared line - and add nothing)
#pragma unroll
      for (int e = 0; e < 8; ++e) in[e] = r[e] >= r_lo && r[e] < r_hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = c_cur[in[e] ? u[e] : 0];
      v4i fa = na, fb = nb;
      v4u fr = nr;
      if (i + 2 < I.steps) {
        fa = __builtin_nontemporal_load(sp + (size_t)(i + 2) * (2 * kPanelThreads));
        fb = __builtin_nontemporal_load(sp + (size_t)(i + 2) * (2 * kPanelThreads) + 1);
        fr = __builtin_nontemporal_load(rp + (size_t)(i + 2) * kPanelThreads);
      }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (in[e] && v[e] != 0.0) atomic_add_noret(&acc[r[e]], v[e]);
      ia = na;
      ib = nb;
      rx = nr;
      na = fa;
      nb = fb;
      nr = fr;
    }
    __syncthreads();
    const uint32_t taken = s_take;  // (written before the barrier above; the next write comes behind the barrier below)
#pragma unroll 4
    for (uint32_t r = tid; r < kPanelRows; r += kPanelThreads)
      if (r >= r_lo && r < r_hi) part[(size_t)I.part0 + r] = acc[r];
    __syncthreads();  // (the accumulators are read: the next item clears them)
    it = item_lo + gridDim.x + taken;
  }
}

// The hub panels' parts: a panel of S > kFoldMin parts has them added kFoldParts at a time first - workgroup (x, g)
// adds the parts [g kFoldParts, ...) of 256 rows, the loads of a lane independent of one another - so that k_dense_apply
// adds ceil(S / kFoldParts) values per row instead of hundreds in a chain.  Rows outside [j_lo, j_hi) are left alone.
__global__ __launch_bounds__(256) void k_panel_fold(double* __restrict__ part, const PanelDesc* __restrict__ panels,
                                                    uint32_t panel, uint32_t j_lo, uint32_t j_hi, const int* state_in) {
  if (dense_state(state_in, kGsJacobi) == kGsNone) return;
  panel += blockIdx.z;
  const PanelDesc P = panels[panel];
  const uint32_t r = blockIdx.x * 256u + threadIdx.x, g = blockIdx.y, j = panel * kPanelRows + r;
  if (P.fold == kNoFold || r >= P.rows || j < j_lo || j >= j_hi) return;
  const uint32_t k0 = g * kFoldParts, k1 = min(P.parts, k0 + kFoldParts);
  if (k0 >= P.parts) return;
  const double* p = part + (size_t)P.base + r;
  double x[kFoldParts];
#pragma unroll
  for (uint32_t k = 0; k < kFoldParts; ++k) x[k] = k0 + k < k1 ? p[(size_t)(k0 + k) * P.rows] : 0.0;
  double v = 0.0;
#pragma unroll
  for (uint32_t k = 0; k < kFoldParts; ++k) v += x[k];
  part[(size_t)P.fold + (size_t)g * P.rows + r] = v;
}

// k_dense_apply: one thread per non-empty row (plus one for a source without in-edges, which
// only ever receives returned dead-end mass): lands the row sum, detects the threshold crossing
// and prepares the row for the next level in place.
// PANEL: the row sum arrives as the S parts k_dense_edges_panel's items left (acc_nz = their buffer; panels = PanelDesc).
template <int MODE, bool PANEL>
__global__ __launch_bounds__(256) void k_dense_apply(const int32_t* __restrict__ nz_rows, uint32_t j_lo, uint32_t n_nz,
                                                      double* __restrict__ acc_nz, const PanelDesc* __restrict__ panels,
                                                      const uint32_t* __restrict__ out_rp,
                                                      const uint32_t* __restrict__ in_rp,
                                                      double* __restrict__ c_cur, double* __restrict__ c_next,
                                                      double* __restrict__ res,
                                                      double* __restrict__ reserve, uint8_t* __restrict__ flags,
                                                      uint32_t* __restrict__ armed,
                                                      DevCounters* ctr, unsigned long long* __restrict__ blk_pack,
                                                      double* __restrict__ blk_dead, uint32_t* __restrict__ blk_ndead,
                                                      int dead_slot, int src_extra, PushArgs a,
                                                      const int* state_in, int state0, int last_block) {
  // rows [j_lo, n_nz) of one block (n_nz = the block's end; + the source without in-edges behind the last block)
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  const int state = dense_state(state_in, state0);
  if (state == kGsNone) return;
  const int tid = threadIdx.x;
  const uint32_t j = j_lo + blockIdx.x * 256u + tid;
  bool have = false;
  int32_t u = -1;
  double acc = 0.0;
  if (j < n_nz) {
    u = nz_rows[j];
    if (PANEL) {
      const PanelDesc P = panels[j / kPanelRows];  // (a wave's rows lie in one panel or two: uniform loads)
      const bool folded = P.fold != kNoFold;
      const double* p = acc_nz + (size_t)(folded ? P.fold : P.base) + (j % kPanelRows);
      const uint32_t cnt = folded ? (P.parts + kFoldParts - 1) / kFoldParts : P.parts;
      uint32_t k = 0;
      for (; k + 4 <= cnt; k += 4) {
        const double x0 = p[(size_t)k * P.rows], x1 = p[(size_t)(k + 1) * P.rows], x2 = p[(size_t)(k + 2) * P.rows],
                     x3 = p[(size_t)(k + 3) * P.rows];
        acc = (((acc + x0) + x1) + x2) + x3;
      }
      for (; k < cnt; ++k) acc += p[(size_t)k * P.rows];
    } else {
      acc = acc_nz[j];
      acc_nz[j] = 0.0;
    }
    have = true;
  } else if (j == n_nz && src_extra) {
    u = a.src;
    have = true;
  }
  double dead_next = 0.0;
  unsigned long long pack = 0, ndead = 0;
  if (have) {
    if (MODE != kBackward && u == a.src) {
      const double dd = ctr->dead[dead_slot];
      if (dd > 0.0) {
        acc += dd;
        ctr->dead[dead_slot] = 0.0;
      }
    }
    const uint32_t d = out_rp[u + 1] - out_rp[u];
    double cn = 0.0;
    if (MODE == kBackward) {
      // Backward_Search.java:73-96 in pull form over the out-CSR: the row's out-neighbours gave (1 - alpha) *
      // residue each, this row takes its share 1 / d_out; strict un-normalised threshold (:89)
      if (acc > 0.0) {
        const double old = res[u];
        const double nw = old + acc / (double)d;
        if (!(old > a.rmax) && nw > a.rmax) {
          reserve[u] = reserve[u] + nw * a.alpha;
          if (old != 0.0) res[u] = 0.0;
          cn = (1.0 - a.alpha) * nw;
          pack = (1ull << kPackShift) | (unsigned long long)(in_rp[u + 1] - in_rp[u]);
        } else {
          res[u] = nw;
        }
      }
    } else if (acc > 0.0) {
      const double old = res[u];
      const double nw = old + acc;
      bool crossing = (MODE == kPower) ? true : (!active_fwd(old, d, a.rmax) && active_fwd(nw, d, a.rmax));
      if (MODE == kFwdTopk) {
        // a row is applied once per level, so "receives mass and meets the threshold" needs no queue test here: an
        // armed node (met the threshold at round start, not queued) joins with its first mass (:226-231)
        if (a.rmax < a.min_rmax && active_fwd(old, d, a.rmax)) crossing = take_armed(armed, u);
        if (active_fwd(nw, d, a.min_rmax)) flags[u] = 1;
      }
      if (crossing) {  // becomes a frontier node of the next level: prepare it right here
        reserve[u] = reserve[u] + nw * a.alpha;
        if (old != 0.0) res[u] = 0.0;
        if (d == 0) {
          dead_next = nw * (1.0 - a.alpha);
          ndead = 1;
        } else {
          cn = ((1.0 - a.alpha) * nw) / (double)d;
        }
        pack = (1ull << kPackShift) | (unsigned long long)d;
      } else {
        res[u] = nw;
      }
    }
    c_next[u] = cn;
    // what the later blocks of this sweep read from the current array (engine.hpp: GsState); nobody reads the last
    // block's rows again in this sweep, and the next sweep reads c_next
    if (!last_block) {
      if (state == kGsEntry) c_cur[u] = c_cur[u] + cn;
      else if (state == kGsInPlace) c_cur[u] = cn;
      else if (state == kGsFlush) c_cur[u] = 0.0;
    }
  }
  // per-workgroup partials; k_dense_reduce sums them (no same-address atomics in this kernel)
  const double ds = block_sum_f64(dead_next, s_red);
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  if (tid == 0) {
    blk_pack[blockIdx.x] = ps;
    blk_dead[blockIdx.x] = ds;
    blk_ndead[blockIdx.x] = (uint32_t)nd;
  }
}

// k_dense_apply_batch: the batched form of k_dense_apply.  Rows without in-edges that can be a query's source
// (those with out-edges) are included: their contribution for the next level is written as 0, or holds the source's
// returned dead-end mass.  So the sweep rewrites every entry of c8_next that can ever be non-zero (isolated nodes'
// entries are never written and stay zero) and a column a slot has left stays all-zero.
// A workgroup takes 64 rows at a time through an LDS tile [row][slot]: row sums come in and next
// contributions go out in the interleaved layout (whole 128-byte lines), while the per-slot
// residue / reserve vectors are walked with a lane per row, i.e. coalesced as in the single-query
// kernel.  Wave w serves kSlotsPerWave consecutive slots; slot arguments are
// wave-uniform.  Counters go to per-slot partials.
constexpr int kApplyRows = 64;
constexpr int kApplyGroups = 2;  // tiles of kApplyRows rows a workgroup carries through its phases together
constexpr int kApplyThreads = 512;  // 8 waves, 2 slots each: few enough slot arguments to stay in SGPRs
constexpr int kSlotsPerWave = kBatch / (kApplyThreads / 64);

__global__ __launch_bounds__(kApplyThreads) void k_dense_apply_batch(const int32_t* __restrict__ nz_rows, uint32_t n_nz,
                                                            const int32_t* __restrict__ zin_rows, uint32_t n_zin,
                                                            double* __restrict__ acc8,
                                                            const uint32_t* __restrict__ out_rp,
                                                            const uint32_t* __restrict__ in_rp_bwd,
                                                            double* __restrict__ c8_cur, double* __restrict__ c8_next,
                                                            uint32_t tile_lo, uint32_t tile_hi, uint32_t gs_mask,
                                                            uint32_t entry_mask,
                                                            const SlotArgs* __restrict__ slots,
                                                            const unsigned long long* __restrict__ cross_bits,
                                                            unsigned long long* __restrict__ prep_bits,
                                                            unsigned long long* __restrict__ blk_pack8,
                                                            double* __restrict__ blk_dead8,
                                                            uint32_t* __restrict__ blk_ndead8, uint32_t part_base,
                                                            uint32_t part_stride) {
  // tiles [tile_lo, tile_hi) of one block of the sweep.  gs_mask: slots whose state writes the current array in place
  // (entry / in-place / flush, engine.hpp: GsState); entry_mask: those of them that add to what it holds.
  __shared__ double tile[kApplyGroups][kApplyRows][kBatch + 1];
  __shared__ double tile_p[kApplyGroups][kApplyRows][kBatch + 1];  // what the rows leave in the current array (slots in gs_mask)
  __shared__ int32_t s_u[kApplyGroups][kApplyRows];
  __shared__ uint32_t s_d[kApplyGroups][kApplyRows];
  __shared__ uint32_t s_din[kApplyGroups][kApplyRows];  // backward sweeps: in-degree = edges the row pushes when it is popped
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform: the slot arguments load into SGPRs
  const uint32_t n_rows = n_nz + n_zin;
  const uint32_t n_tiles = (n_rows + kApplyRows - 1) / kApplyRows;
  SlotArgs a[kSlotsPerWave];
#pragma unroll
  for (int i = 0; i < kSlotsPerWave; ++i) a[i] = slots[w * kSlotsPerWave + i];
  double dead_next[kSlotsPerWave];
  unsigned long long pack[kSlotsPerWave];
  uint32_t ndead[kSlotsPerWave];
#pragma unroll
  for (int i = 0; i < kSlotsPerWave; ++i) {
    dead_next[i] = 0.0;
    pack[i] = 0;
    ndead[i] = 0;
  }
  // kApplyGroups tiles of 64 rows per trip: every phase below issues the loads of all of them before the barrier that
  // ends it (the waves of a workgroup spent four fifths of their cycles at those barriers with one tile per trip)
  for (uint32_t tl0 = tile_lo + blockIdx.x * kApplyGroups; tl0 < tile_hi; tl0 += gridDim.x * kApplyGroups) {
#pragma unroll
    for (int g = 0; g < kApplyGroups; ++g) {
      const uint32_t tl = tl0 + g;
      const bool in = tl < tile_hi;
      const uint32_t row0 = tl * kApplyRows;
      // rows inside one 512-edge chunk are rewritten by plain stores every sweep; only the rows that
      // cross a chunk boundary are summed with atomics and have to be cleared for the next sweep
      const unsigned long long cw = in ? cross_bits[tl] : 0ull;
#pragma unroll
      for (int i = 0; i < kApplyRows * kBatch / kApplyThreads; ++i) {
        const uint32_t idx = (uint32_t)i * (uint32_t)kApplyThreads + tid;
        const uint32_t r = idx / kBatch, s = idx % kBatch;
        const uint32_t j = row0 + r;
        double v = 0.0;
        if (in && j < n_nz) {
          const size_t t = (size_t)j * kBatch + s;
          v = __builtin_nontemporal_load(&acc8[t]);
          if (v != 0.0 && ((cw >> r) & 1ull)) acc8[t] = 0.0;
        }
        tile[g][r][s] = v;
      }
    }
    if (tid < kApplyRows * kApplyGroups) {
      const uint32_t g = tid / kApplyRows, r = tid % kApplyRows;
      const uint32_t tl = tl0 + g;
      const uint32_t j = tl * kApplyRows + r;
      const int32_t u = tl >= tile_hi ? -1 : (j < n_nz ? nz_rows[j] : (j < n_rows ? zin_rows[j - n_nz] : -1));
      s_u[g][r] = u;
      s_d[g][r] = u >= 0 ? out_rp[u + 1] - out_rp[u] : 0u;
      s_din[g][r] = (u >= 0 && in_rp_bwd) ? in_rp_bwd[u + 1] - in_rp_bwd[u] : 0u;
    }
    __syncthreads();
    if (entry_mask) {  // entry sweeps add to the row's own pending contribution: stage it (whole lines)
#pragma unroll
      for (int g = 0; g < kApplyGroups; ++g) {
#pragma unroll
        for (int i = 0; i < kApplyRows * kBatch / kApplyThreads; ++i) {
          const uint32_t idx = (uint32_t)i * (uint32_t)kApplyThreads + tid;
          const uint32_t r = idx / kBatch, s = idx % kBatch;
          const int32_t ur = s_u[g][r];
          tile_p[g][r][s] = (ur >= 0 && (entry_mask >> s & 1u)) ? c8_cur[(size_t)ur * kBatch + s] : 0.0;
        }
      }
      __syncthreads();
    }
    // the wave's slots in three passes, so that all their residue / reserve loads are in flight together:
    // (1) row sums (+ the source's returned dead-end mass), (2) loads, (3) arithmetic and stores
    int32_t u[kApplyGroups];
    uint32_t d[kApplyGroups], din[kApplyGroups];
    double accv[kApplyGroups][kSlotsPerWave], oldv[kApplyGroups][kSlotsPerWave], rsvv[kApplyGroups][kSlotsPerWave];
    bool live[kApplyGroups][kSlotsPerWave];
#pragma unroll
    for (int g = 0; g < kApplyGroups; ++g) {
      u[g] = s_u[g][lane];
      d[g] = s_d[g][lane];
      din[g] = s_din[g][lane];
#pragma unroll
      for (int i = 0; i < kSlotsPerWave; ++i) {
        double acc = tile[g][lane][w * kSlotsPerWave + i];
        const bool on = a[i].active && u[g] >= 0;
        if (on && a[i].mode != kBackward && u[g] == a[i].src) {
          const double dd = a[i].ctr->dead[a[i].dead_slot];
          if (dd > 0.0) {
            acc += dd;
            a[i].ctr->dead[a[i].dead_slot] = 0.0;
          }
        }
        accv[g][i] = acc;
        live[g][i] = on && acc > 0.0;
      }
    }
#pragma unroll
    for (int g = 0; g < kApplyGroups; ++g) {
#pragma unroll
      for (int i = 0; i < kSlotsPerWave; ++i) {
        oldv[g][i] = live[g][i] ? a[i].res[u[g]] : 0.0;
        rsvv[g][i] = live[g][i] ? a[i].reserve[u[g]] : 0.0;  // needed when the row crosses, which most rows of a dense level do
      }
    }
#pragma unroll
    for (int g = 0; g < kApplyGroups; ++g) {
#pragma unroll
      for (int i = 0; i < kSlotsPerWave; ++i) {
        const int s = w * kSlotsPerWave + i;
        double cn = 0.0;
        if (live[g][i] && a[i].mode == kBackward) {
          // Backward_Search.java:73-96 in pull form: the row's out-neighbours' (1 - alpha) * residue, divided by
          // this row's out-degree; strict un-normalised threshold
          const double old = oldv[g][i];
          const double nw = old + accv[g][i] / (double)d[g];
          if (!(old > a[i].rmax) && nw > a[i].rmax) {
            a[i].reserve[u[g]] = rsvv[g][i] + nw * a[i].alpha;
            if (oldv[g][i] != 0.0) a[i].res[u[g]] = 0.0;  // (rows that cross every sweep hold zero already)
            cn = (1.0 - a[i].alpha) * nw;
            pack[i] += (1ull << kPackShift) | (unsigned long long)din[g];
          } else {
            a[i].res[u[g]] = nw;
          }
        } else if (live[g][i]) {
          const double old = oldv[g][i];
          const double nw = old + accv[g][i];
          bool crossing = !active_fwd(old, d[g], a[i].rmax) && active_fwd(nw, d[g], a[i].rmax);
          if (a[i].mode == kFwdTopk) {
            if (a[i].rmax < a[i].min_rmax && active_fwd(old, d[g], a[i].rmax)) crossing = take_armed(a[i].armed, u[g]);
            if (active_fwd(nw, d[g], a[i].min_rmax)) a[i].flags[u[g]] = 1;
          }
          if (crossing) {  // becomes a frontier node of the next level: prepare it right here
            a[i].reserve[u[g]] = rsvv[g][i] + nw * a[i].alpha;
            if (oldv[g][i] != 0.0) a[i].res[u[g]] = 0.0;  // (rows that cross every sweep hold zero already)
            if (d[g] == 0) {
              dead_next[i] += nw * (1.0 - a[i].alpha);
              ndead[i]++;
            } else {
              cn = ((1.0 - a[i].alpha) * nw) / (double)d[g];
            }
            pack[i] += (1ull << kPackShift) | (unsigned long long)d[g];
          } else {
            a[i].res[u[g]] = nw;
          }
        }
        tile[g][lane][s] = cn;
        if (gs_mask >> s & 1u) {
          const int gst = a[i].gs_state;
          // entry: old + new (the old value was staged in tile_p above)
          tile_p[g][lane][s] = gst == kGsEntry ? tile_p[g][lane][s] + cn : (gst == kGsInPlace ? cn : 0.0);
        }
        // rows of this tile that hold a contribution for the slot's next level (read when the slot
        // goes back to list form)
        const unsigned long long bits = __ballot(cn > 0.0);
        if (lane == 0 && a[i].active && tl0 + g < tile_hi) prep_bits[(size_t)s * n_tiles + tl0 + g] = bits;
      }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kApplyGroups; ++g) {
#pragma unroll
      for (int i = 0; i < kApplyRows * kBatch / kApplyThreads; ++i) {
        const uint32_t idx = (uint32_t)i * (uint32_t)kApplyThreads + tid;
        const uint32_t r = idx / kBatch, s = idx % kBatch;
        const int32_t ur = s_u[g][r];
        if (ur >= 0) {
          c8_next[(size_t)ur * kBatch + s] = tile[g][r][s];
          if (gs_mask >> s & 1u) c8_cur[(size_t)ur * kBatch + s] = tile_p[g][r][s];
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < kSlotsPerWave; ++i) {
    const double ds = wave_sum_f64(dead_next[i]);
    const unsigned long long ps = wave_sum_u64(pack[i]);
    const unsigned long long nd = wave_sum_u64((unsigned long long)ndead[i]);
    if (lane == 0) {
      const size_t o = (size_t)(w * kSlotsPerWave + i) * part_stride + part_base + blockIdx.x;
      blk_pack8[o] = ps;
      blk_dead8[o] = ds;
      blk_ndead8[o] = (uint32_t)nd;
    }
  }
}

// workgroup s sums slot s's partials into that slot's counters
__global__ __launch_bounds__(1024) void k_dense_reduce_batch(const unsigned long long* __restrict__ blk_pack8,
                                                           const double* __restrict__ blk_dead8,
                                                           const uint32_t* __restrict__ blk_ndead8, uint32_t n_blocks,
                                                           uint32_t part_stride,
                                                           const SlotArgs* __restrict__ slots,
                                                           unsigned long long* __restrict__ sweep_out) {
  __shared__ double s_red[16];
  __shared__ unsigned long long s_red2[16];
  const SlotArgs a = slots[blockIdx.x];
  if (!a.active) return;
  unsigned long long pack = 0, ndead = 0;
  double dead = 0.0;
  for (uint32_t i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    pack += blk_pack8[(size_t)blockIdx.x * part_stride + i];
    dead += blk_dead8[(size_t)blockIdx.x * part_stride + i];
    ndead += blk_ndead8[(size_t)blockIdx.x * part_stride + i];
  }
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  const double ds = block_sum_f64(dead, s_red);
  if (threadIdx.x == 0) {
    a.ctr->packed[a.out_slot] = ps;
    sweep_out[blockIdx.x] = ps;  // all slots' new frontier counters side by side: one read-back per sweep
    if (nd) {
      a.ctr->dead[a.dead_slot ^ 1] = a.ctr->dead[a.dead_slot ^ 1] + ds;
      a.ctr->dead_pops += nd;
    }
  }
}

// sums the per-workgroup partials of a dense level into the level counter, the dead-mass cell
// and the dead-end pop count
__global__ __launch_bounds__(1024) void k_dense_reduce(const unsigned long long* __restrict__ blk_pack,
                                                        const double* __restrict__ blk_dead,
                                                        const uint32_t* __restrict__ blk_ndead, uint32_t n_blocks,
                                                        DevCounters* ctr, int out_slot, int dead_slot_next,
                                                        const int* state_in, int state0, unsigned long long* hist_out,
                                                        int* state_out, unsigned long long dense_thresh,
                                                        unsigned long long gs_thresh, uint32_t* queues) {
  __shared__ double s_red[16];
  __shared__ unsigned long long s_red2[16];
  if (queues && threadIdx.x < kPanelQueues) queues[threadIdx.x] = 0u;  // the item queues of this level's panel launches
  const int state = dense_state(state_in, state0);
  if (state == kGsNone) {
    if (threadIdx.x == 0 && state_out) *state_out = kGsNone;
    return;
  }
  unsigned long long pack = 0, ndead = 0;
  double dead = 0.0;
  for (uint32_t i = threadIdx.x; i < n_blocks; i += blockDim.x) {
    pack += blk_pack[i];
    if (blk_dead) {
      dead += blk_dead[i];
      ndead += blk_ndead[i];
    }
  }
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  const double ds = block_sum_f64(dead, s_red);
  if (threadIdx.x == 0) {
    ctr->packed[out_slot] = ps;
    if (hist_out) *hist_out = ps;
    if (state_out) *state_out = gs_next_state(state, ps >> kPackShift, ps & kPackMask, dense_thresh, gs_thresh);
    if (nd) {
      ctr->dead[dead_slot_next] = ctr->dead[dead_slot_next] + ds;
      ctr->dead_pops += nd;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// frontier seeding (round starts) and conversions between the two level shapes
// ------------------------------------------------------------------------------------------------
// seed kinds: 0 = every node that meets the (new) threshold (a FORA round after a halving);
//             1 = top-k round start from the parked set (Forward_Push.java:163,173,241-247)
template <int KIND>
__device__ __forceinline__ bool seed_pred(uint32_t v, const double* __restrict__ res, uint32_t d,
                                          const uint8_t* __restrict__ flags, const PushArgs& a) {
  if (KIND == 1 && !flags[v]) return false;
  return active_fwd(res[v], d, a.rmax);
}

// top-k round start: parked nodes that start the round leave the parked set; parked nodes that
// fell below min_rmax are dropped (Forward_Push.java:241-247 keeps the others parked)
__device__ __forceinline__ void unpark(uint32_t v, const double* __restrict__ res, uint32_t d,
                                       uint8_t* __restrict__ flags, const PushArgs& a) {
  if (!flags[v]) return;
  const double r = res[v];
  if (active_fwd(r, d, a.rmax) || !active_fwd(r, d, a.min_rmax)) flags[v] = 0;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_count_active(uint32_t n, const double* __restrict__ res,
                                                       const uint32_t* __restrict__ out_rp,
                                                       const uint8_t* __restrict__ flags, uint32_t* __restrict__ armed,
                                                       unsigned long long* __restrict__ blk_pack,
                                                       unsigned long long* zero_word, PushArgs a) {
  __shared__ unsigned long long s_red2[4];
  unsigned long long pack = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0ull;  // the list counter of the seeding pass that follows
  // wave-uniform trip count: a wave covers 64 consecutive nodes, whose "armed" bits it writes as one 64-bit word
  for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += gridDim.x * blockDim.x) {
    const uint32_t v = base + (threadIdx.x & 63u);
    bool arm = false;
    if (v < n) {
      const uint32_t d = out_rp[v + 1] - out_rp[v];
      if (seed_pred<KIND>(v, res, d, flags, a)) pack += (1ull << kPackShift) | (unsigned long long)d;
      // top-k rounds: meets the round's threshold without being parked (only when rmax < min_rmax)
      if (KIND == 1) arm = a.rmax < a.min_rmax && !flags[v] && active_fwd(res[v], d, a.rmax);
    }
    if (KIND == 1) {
      const unsigned long long bits = __ballot(arm);
      if ((threadIdx.x & 63u) == 0) {
        armed[base >> 5] = (uint32_t)bits;
        armed[(base >> 5) + 1] = (uint32_t)(bits >> 32);
      }
    }
  }
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  if (threadIdx.x == 0) blk_pack[blockIdx.x] = ps;
}

// `armed` (top-k round starts that do not count first): the pass also writes the round's "armed" bits, as
// k_count_active does; the workgroups' ranges are multiples of 256 nodes then, so a wave covers 64 consecutive ids.
template <int KIND>
__global__ __launch_bounds__(256) void k_seed_list(uint32_t n, const double* __restrict__ res,
                                                    const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                    int32_t* __restrict__ Fn, uint32_t* __restrict__ eoffn,
                                                    unsigned long long* counter, uint32_t* __restrict__ armed, PushArgs a) {
  uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  if (armed) per = (per + 255u) & ~255u;
  const unsigned long long lo64 = (unsigned long long)blockIdx.x * per;
  if (lo64 >= n) return;
  const uint32_t lo = (uint32_t)lo64;
  const uint32_t hi = lo64 + per < n ? lo + per : n;
  block_range_compact(
      lo, hi, counter,
      [&](uint32_t v, unsigned long long* w) {
        const uint32_t d = out_rp[v + 1] - out_rp[v];
        *w = d;
        return seed_pred<KIND>(v, res, d, flags, a);
      },
      [&](uint32_t v, uint32_t pos, unsigned long long eo, unsigned long long) {
        Fn[pos] = (int32_t)v;
        eoffn[pos] = (uint32_t)eo;
      });
  if (KIND == 1) {
    __syncthreads();
    // wave-uniform trip count: a wave covers 64 consecutive nodes, whose "armed" bits it writes as one 64-bit word
    for (uint32_t base = lo + (threadIdx.x & ~63u); base < hi; base += 256) {
      const uint32_t v = base + (threadIdx.x & 63u);
      bool arm = false;
      if (v < hi) {
        const uint32_t d = out_rp[v + 1] - out_rp[v];
        // meets the round's threshold without being parked (only when rmax < min_rmax); tested before the node leaves
        // the parked set
        if (armed) arm = a.rmax < a.min_rmax && !flags[v] && active_fwd(res[v], d, a.rmax);
        unpark(v, res, d, flags, a);
      }
      if (armed) {
        const unsigned long long bits = __ballot(arm);
        if ((threadIdx.x & 63u) == 0) {
          armed[base >> 5] = (uint32_t)bits;
          armed[(base >> 5) + 1] = (uint32_t)(bits >> 32);
        }
      }
    }
  }
}

// k_seed_list<1> in one pass (round 5): a workgroup takes tiles of 2048 consecutive nodes, a thread 8 consecutive ones -
// its flags are one 8-byte load, its residues four 16-byte loads, its row pointers three - lists the round's start set
// (block_tile_compact), writes the armed bits of its 8 nodes as one byte and lets the parked nodes go with one 8-byte
// store.  Same list order (ascending ids), same bits, same flags as the two-pass kernel; 41 -> ~10 us on R-MAT 22.
constexpr int kSeedItems = 8;
__global__ __launch_bounds__(256) void k_seed_list_topk(uint32_t n, const double* __restrict__ res,
                                                         const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                         int32_t* __restrict__ Fn, uint32_t* __restrict__ eoffn,
                                                         unsigned long long* counter, uint32_t* __restrict__ armed, PushArgs a) {
  const uint32_t tile = 256u * kSeedItems;
  const uint32_t n_tiles = (n + tile - 1) / tile;
  for (uint32_t tl = blockIdx.x; tl < n_tiles; tl += gridDim.x) {
    const uint32_t v0 = tl * tile + threadIdx.x * kSeedItems;
    bool take[kSeedItems];
    unsigned long long w[kSeedItems];
    double r[kSeedItems];
    uint32_t deg[kSeedItems];
    uint8_t fl[kSeedItems];
    if (v0 + kSeedItems <= n) {  // (n + 1 row pointers and n flags exist: whole groups of 8 load as vectors)
      const unsigned long long f8 = *reinterpret_cast<const unsigned long long*>(flags + v0);
#pragma unroll
      for (int i = 0; i < kSeedItems; ++i) fl[i] = (uint8_t)(f8 >> (8 * i));
      const double2* r2 = reinterpret_cast<const double2*>(res + v0);
#pragma unroll
      for (int i = 0; i < kSeedItems / 2; ++i) {
        const double2 x = r2[i];
        r[2 * i] = x.x;
        r[2 * i + 1] = x.y;
      }
      const uint4* p4 = reinterpret_cast<const uint4*>(out_rp + v0);  // (v0 is a multiple of 8: 32-byte aligned)
      const uint4 x0 = p4[0], x1 = p4[1];
      const uint32_t last = out_rp[v0 + 8];
      deg[0] = x0.y - x0.x; deg[1] = x0.z - x0.y; deg[2] = x0.w - x0.z; deg[3] = x1.x - x0.w;
      deg[4] = x1.y - x1.x; deg[5] = x1.z - x1.y; deg[6] = x1.w - x1.z; deg[7] = last - x1.w;
    } else {
#pragma unroll
      for (int i = 0; i < kSeedItems; ++i) {
        const bool in = v0 + i < n;
        fl[i] = in ? flags[v0 + i] : 0;
        r[i] = in ? res[v0 + i] : 0.0;
        deg[i] = in ? out_rp[v0 + i + 1] - out_rp[v0 + i] : 0u;
      }
    }
    uint32_t arm_bits = 0;
    unsigned long long f_new = 0;
#pragma unroll
    for (int i = 0; i < kSeedItems; ++i) {
      const uint32_t d = deg[i];
      const bool in = v0 + i < n;
      const bool act = in && active_fwd(r[i], d, a.rmax);
      take[i] = act && fl[i];  // seed_pred<1>
      w[i] = d;
      // meets the round's threshold without being parked (only when rmax < min_rmax): tested before the node leaves the set
      if (a.rmax < a.min_rmax && !fl[i] && act) arm_bits |= 1u << i;
      // unpark: parked nodes that start the round leave the set; those below min_rmax are dropped
      uint8_t f = fl[i];
      if (f && (act || !active_fwd(r[i], d, a.min_rmax))) f = 0;
      f_new |= (unsigned long long)f << (8 * i);
    }
    block_tile_compact<kSeedItems>(take, w, counter, [&](int i, uint32_t pos, unsigned long long eo) {
      Fn[pos] = (int32_t)(v0 + i);
      eoffn[pos] = (uint32_t)eo;
    });
    if (v0 < n) {
      if (armed) reinterpret_cast<uint8_t*>(armed)[v0 >> 3] = (uint8_t)arm_bits;
      if (v0 + kSeedItems <= n) {
        *reinterpret_cast<unsigned long long*>(flags + v0) = f_new;
      } else {
#pragma unroll
        for (int i = 0; i < kSeedItems; ++i)
          if (v0 + i < n) flags[v0 + i] = (uint8_t)(f_new >> (8 * i));
      }
    }
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_seed_dense(uint32_t n, double* __restrict__ res, double* __restrict__ reserve,
                                                     const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                     CView c_dense, unsigned long long* __restrict__ blk_pack,
                                                     double* __restrict__ blk_dead, uint32_t* __restrict__ blk_ndead,
                                                     PushArgs a) {
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  double dead = 0.0;
  unsigned long long pack = 0, ndead = 0;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t d = out_rp[v + 1] - out_rp[v];
    double c = 0.0;
    const bool take = seed_pred<KIND>(v, res, d, flags, a);
    if (KIND == 1) unpark(v, res, d, flags, a);
    if (take) {
      const double rc = res[v];
      res[v] = 0.0;
      reserve[v] = reserve[v] + rc * a.alpha;
      if (d == 0) {
        dead += rc * (1.0 - a.alpha);
        ndead++;
      } else {
        c = ((1.0 - a.alpha) * rc) / (double)d;
      }
      pack += (1ull << kPackShift) | (unsigned long long)d;
    }
    if (c_dense.stride == 1 || c != 0.0) c_dense.at(v) = c;  // a slot's column is all-zero beforehand
  }
  const double ds = block_sum_f64(dead, s_red);
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  if (threadIdx.x == 0) {
    blk_pack[blockIdx.x] = ps;
    blk_dead[blockIdx.x] = ds;
    blk_ndead[blockIdx.x] = (uint32_t)nd;
  }
}

// dense-prepared state -> sparse-prepared state: list every node holding a contribution
__global__ __launch_bounds__(256) void k_compact_prepared(uint32_t n, CView c_dense, bool clear,
                                                           const uint32_t* __restrict__ trp, int32_t* __restrict__ Fn,
                                                           uint32_t* __restrict__ eoffn, double* __restrict__ cF,
                                                           unsigned long long* counter) {
  const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
  const uint32_t lo = blockIdx.x * per;
  const uint32_t hi = lo + per < n ? lo + per : n;
  if (lo >= hi) return;
  block_range_compact(
      lo, hi, counter,
      [&](uint32_t v, unsigned long long* w) {
        if (!(c_dense.at(v) > 0.0)) return false;
        *w = trp[v + 1] - trp[v];
        return true;
      },
      [&](uint32_t v, uint32_t pos, unsigned long long eo, unsigned long long) {
        Fn[pos] = (int32_t)v;
        eoffn[pos] = (uint32_t)eo;
        cF[pos] = c_dense.at(v);
        if (clear) c_dense.at(v) = 0.0;  // a slot leaving the dense shape hands back an all-zero column
      });
}

// the same for a batch slot after a sweep: the apply kernel left one bit per row ordinal that holds a
// contribution, so only those entries of the slot's column are read (and handed back as zero)
__global__ __launch_bounds__(256) void k_compact_bits(uint32_t n_rows, uint32_t n_nz,
                                                       const unsigned long long* __restrict__ bits,
                                                       const int32_t* __restrict__ nz_rows,
                                                       const int32_t* __restrict__ zin_rows, CView c_dense,
                                                       const uint32_t* __restrict__ trp, int32_t* __restrict__ Fn,
                                                       uint32_t* __restrict__ eoffn, double* __restrict__ cF,
                                                       unsigned long long* counter) {
  const uint32_t per = (((n_rows + gridDim.x - 1) / gridDim.x) + 63u) & ~63u;
  const uint32_t lo = blockIdx.x * per;
  const uint32_t hi = lo + per < n_rows ? lo + per : n_rows;
  if (lo >= hi) return;
  block_range_compact(
      lo, hi, counter,
      [&](uint32_t j, unsigned long long* w) {
        if (!((bits[j >> 6] >> (j & 63)) & 1ull)) return false;
        const int32_t u = j < n_nz ? nz_rows[j] : zin_rows[j - n_nz];
        *w = trp[u + 1] - trp[u];
        return true;
      },
      [&](uint32_t j, uint32_t pos, unsigned long long eo, unsigned long long) {
        const int32_t u = j < n_nz ? nz_rows[j] : zin_rows[j - n_nz];
        Fn[pos] = u;
        eoffn[pos] = (uint32_t)eo;
        cF[pos] = c_dense.at((uint32_t)u);
        c_dense.at((uint32_t)u) = 0.0;
      });
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_partial(const double* __restrict__ x, uint32_t n,
                                                      double* __restrict__ partial) {
  __shared__ double s_red[4];
  double acc = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) acc += x[i];
  const double s = block_sum_f64(acc, s_red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_sum_final(const double* __restrict__ partial, uint32_t np, DevCounters* ctr) {
  __shared__ double s_red[4];
  double acc = 0.0;
  for (uint32_t i = threadIdx.x; i < np; i += blockDim.x) acc += partial[i];
  const double s = block_sum_f64(acc, s_red);
  if (threadIdx.x == 0) ctr->sum_out = s;
}

__global__ void k_set_f64(double* p, uint32_t idx, double value) { p[idx] = value; }

__global__ __launch_bounds__(256) void k_permute_out(const double* __restrict__ x, const int32_t* __restrict__ old2new,
                                                      double* __restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = x[old2new[i]];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t work, uint32_t per_block, uint32_t cap) {
  uint64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (uint32_t)b;
}

// contribution buffer `cbuf` of a handle: its own array, or its column of the parent's c8 array
static inline CView cview(pprhip_graph* g, int cbuf) {
  if (g->parent) return CView{g->parent->c8[cbuf], (uint32_t)kBatch, (uint32_t)g->slot_index};
  return CView{g->cdense[cbuf], 1u, 0u};
}

#define DISPATCH_MODE(MODEVAR, ...)                                       \
  switch (MODEVAR) {                                                      \
    case kFwdWhole: { constexpr int M = kFwdWhole; __VA_ARGS__; } break;  \
    case kFwdTopk: { constexpr int M = kFwdTopk; __VA_ARGS__; } break;    \
    case kBackward: { constexpr int M = kBackward; __VA_ARGS__; } break;  \
    default: { constexpr int M = kPower; __VA_ARGS__; } break;            \
  }

int launch_sparse_prepare(pprhip_graph* g, const PushArgs& a, int fbuf, int level, uint64_t nf_upper,
                          unsigned long long dense_thresh, bool scatter_dense, int cbuf, int dead_slot,
                          unsigned long long pk0) {
  const uint32_t grid = grid_for(nf_upper, 256, 512);
  const CView cd = scatter_dense ? cview(g, cbuf) : CView{nullptr, 1, 0};
  DISPATCH_MODE(a.mode, k_sparse_prepare<M><<<dim3(grid), dim3(256), 0, g->stream>>>(
                            g->F[fbuf], g->out_rp, g->residue, g->reserve, g->cF, cd, g->ctr, level, dense_thresh,
                            dead_slot, pk0, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_sparse_push(pprhip_graph* g, const PushArgs& a, int fbuf, int level, uint64_t ef_upper,
                       unsigned long long dense_thresh, int dead_slot, unsigned long long pk0) {
  const uint32_t grid = grid_for(ef_upper, kPushTile, 2048);
  const bool bwd = a.mode == kBackward;
  const uint32_t* trp = bwd ? g->in_rp : g->out_rp;
  const int32_t* tci = bwd ? g->in_ci : g->out_ci;
  // (the override exists for the tests, which run the table on graphs far below the default switch-over)
  const char* comb_env = hook_env("PPRHIP_COMB_MIN_EDGES");
  const unsigned long long comb_min = comb_env ? strtoull(comb_env, nullptr, 10) : (unsigned long long)kCombMinEdges;
  DISPATCH_MODE(a.mode, k_sparse_push<M><<<dim3(grid), dim3(256), 0, g->stream>>>(
                            g->F[fbuf], g->cF, g->eoff[fbuf], trp, tci, g->out_ext, g->in_rp, g->residue, g->flags,
                            g->armed, g->F[fbuf ^ 1], g->eoff[fbuf ^ 1], g->ctr, level, dense_thresh, dead_slot,
                            comb_min, pk0, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_sparse_levels_wg(pprhip_graph* g, const PushArgs& a, int fbuf0, int first, int last,
                            unsigned long long dense_thresh, unsigned long long wg_cap, int dead_slot,
                            unsigned long long pk0) {
  const bool bwd = a.mode == kBackward;
  const uint32_t* trp = bwd ? g->in_rp : g->out_rp;
  const int32_t* tci = bwd ? g->in_ci : g->out_ci;
  const char* comb_env = hook_env("PPRHIP_COMB_MIN_EDGES");
  const unsigned long long comb_min = comb_env ? strtoull(comb_env, nullptr, 10) : (unsigned long long)kCombMinEdges;
  DISPATCH_MODE(a.mode, k_sparse_levels_wg<M><<<dim3(1), dim3(256), 0, g->stream>>>(
                            g->F[0], g->F[1], g->eoff[0], g->eoff[1], g->out_rp, g->residue, g->reserve, g->cF, trp, tci,
                            g->out_ext, g->in_rp, g->flags, g->armed, g->ctr, fbuf0, first, last, dense_thresh, wg_cap,
                            dead_slot, comb_min, pk0, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_dense_level(pprhip_graph* g, const PushArgs& a, int cbuf, int out_slot, int dead_slot,
                       const DenseLaunch& dl) {
  // forward levels pull over the in-CSR, backward levels over the out-CSR (layout built by the caller)
  const bool bwd = a.mode == kBackward;
  const int32_t* ci = bwd ? g->out_ci : g->in_ci;
  const uint8_t* flags = bwd ? g->start_flags_o : g->start_flags;
  const uint32_t* cstarts = bwd ? g->chunk_starts_o : g->chunk_starts;
  const int32_t* nz = bwd ? g->nz_rows_o : g->nz_rows;
  const uint32_t n_nz = bwd ? g->n_nz_o : g->n_nz;
  // a source without in-edges still receives returned dead-end mass: one extra apply thread, behind the last block
  const int src_extra = (!bwd && a.src >= 0 && g->h_in_rp[a.src + 1] == g->h_in_rp[a.src]) ? 1 : 0;
  const GsBlock whole{0u, n_nz, 0ull, (unsigned long long)g->m};
  const GsBlock* blocks = (dl.blocks && dl.n_blocks > 1 && !bwd) ? dl.blocks : &whole;
  const int nb = blocks == &whole ? 1 : dl.n_blocks;
  const uint32_t n_hot = g->relabeled ? std::min<uint32_t>(g->n, (uint32_t)kHotMax) : 0u;
  // forward sweeps walk the row-panel copy of the in-CSR where the graph has one (from 2^20 edges on), else - a graph
  // whose sources span several slices - the sliced copy
  const PanelLayout* pn = (!bwd && g->pn && g->pn_part) ? g->pn : nullptr;
  const SlicedLayout* sl = (bwd || pn) ? nullptr : g->sl;
  const EdgeWindows* wins = sl ? detail::sliced_windows_of(g, blocks == &whole ? nullptr : blocks, nb) : nullptr;
  if (sl) {
    ci = sl->ci;
    flags = sl->flags;
    cstarts = sl->chunk_starts;
  }
  uint32_t part_base = 0;
  for (int b = 0; b < nb; ++b) {
    const GsBlock& B = blocks[b];
    EdgeWindows one;
    if (!sl) {
      one.n = 1;
      one.c_pre[0] = 0;
      one.c_lo[0] = (uint32_t)(B.e_lo / kChunkEdges);
      one.c_pre[1] = (uint32_t)((B.e_hi + kChunkEdges - 1) / kChunkEdges) - one.c_lo[0];
      one.e_lo[0] = B.e_lo;
      one.e_hi[0] = B.e_hi;
    }
    const EdgeWindows& W = sl ? wins[b] : one;
    const uint32_t n_ch = W.n ? W.c_pre[W.n] : 0u;
    if (pn) {
      // block boundaries are multiples of 256 row ordinals and may cut a panel: the kernel leaves the other rows out
      const uint32_t p_lo = B.j_lo / kPanelRows, p_hi = std::min<uint32_t>(pn->n_panels, (B.j_hi + kPanelRows - 1) / kPanelRows);
      const uint32_t i_lo = p_hi > p_lo ? pn->h_panel_item0[p_lo] : 0u, i_hi = p_hi > p_lo ? pn->h_panel_item0[p_hi] : 0u;
      if (i_hi > i_lo) {
        const uint32_t grid = std::min<uint32_t>(i_hi - i_lo, (uint32_t)g->n_cus * (uint32_t)(160 * 1024 / (kPanelLdsBytes + 1024)));
        k_dense_edges_panel<<<dim3(grid), dim3(kPanelThreads), kPanelLdsBytes, g->stream>>>(
            pn->src, pn->rloc, pn->items, i_lo, i_hi, g->cdense[cbuf], g->pn_part, B.j_lo, B.j_hi, n_nz, dl.state_in,
            g->pn_ctr + std::min(b, kPanelQueues - 1));
        PPRHIP_CHECK_HIP(hipGetLastError());
        // panels of many parts (the hub rows': the first few - rows are ordered by degree, so parts do not grow)
        uint32_t p_fold = p_lo, s_max = 0;
        for (uint32_t p = p_lo; p < p_hi; ++p) {
          const uint32_t S = pn->h_panel_item0[p + 1] - pn->h_panel_item0[p];
          if (S > kFoldMin) {
            p_fold = p + 1;
            s_max = std::max(s_max, S);
          }
        }
        if (p_fold > p_lo) {
          k_panel_fold<<<dim3(kPanelRows / 256, (s_max + kFoldParts - 1) / kFoldParts, p_fold - p_lo), dim3(256), 0, g->stream>>>(
              g->pn_part, pn->panels, p_lo, B.j_lo, B.j_hi, dl.state_in);
          PPRHIP_CHECK_HIP(hipGetLastError());
        }
      }
    } else if (g->n_chunks && n_ch) {
      // persistent workgroups: one 1024-thread workgroup per CU when the LDS hot table is in use
      const uint32_t want = (n_ch + 15) / 16;
      const uint32_t grid = std::min<uint32_t>(want, (uint32_t)g->n_cus * (n_hot ? 1u : 2u));
      const size_t lds = n_hot ? sizeof(double) * n_hot : 0;  // (above 64 KB: opted in by init_kernels_push)
      if (n_hot && sl)
        k_dense_edges<true, true><<<dim3(grid), dim3(1024), lds, g->stream>>>(
            ci, flags, cstarts, sl->seg_row, W, g->cdense[cbuf], g->acc_nz, n_hot, dl.state_in);
      else if (n_hot)
        k_dense_edges<true, false><<<dim3(grid), dim3(1024), lds, g->stream>>>(
            ci, flags, cstarts, nullptr, W, g->cdense[cbuf], g->acc_nz, n_hot, dl.state_in);
      else if (sl)
        k_dense_edges<false, true><<<dim3(grid), dim3(1024), 0, g->stream>>>(
            ci, flags, cstarts, sl->seg_row, W, g->cdense[cbuf], g->acc_nz, 0u, dl.state_in);
      else
        k_dense_edges<false, false><<<dim3(grid), dim3(1024), 0, g->stream>>>(
            ci, flags, cstarts, nullptr, W, g->cdense[cbuf], g->acc_nz, 0u, dl.state_in);
      PPRHIP_CHECK_HIP(hipGetLastError());
    }
    const int extra = (b == nb - 1) ? src_extra : 0;
    const uint32_t rows = B.j_hi - B.j_lo + (uint32_t)extra;
    const uint32_t grid = (rows + 255) / 256;
    if (grid) {
      if (pn) {
        DISPATCH_MODE(a.mode, (k_dense_apply<M, true><<<dim3(grid), dim3(256), 0, g->stream>>>(
                                  nz, B.j_lo, B.j_hi, g->pn_part, pn->panels, g->out_rp, g->in_rp, g->cdense[cbuf],
                                  g->cdense[cbuf ^ 1], g->residue, g->reserve, g->flags, g->armed, g->ctr,
                                  g->blk_pack + part_base, g->blk_dead + part_base, g->blk_ndead + part_base, dead_slot,
                                  extra, a, dl.state_in, dl.state0, b == nb - 1 ? 1 : 0)));
      } else {
        DISPATCH_MODE(a.mode, (k_dense_apply<M, false><<<dim3(grid), dim3(256), 0, g->stream>>>(
                                  nz, B.j_lo, B.j_hi, g->acc_nz, nullptr, g->out_rp, g->in_rp, g->cdense[cbuf],
                                  g->cdense[cbuf ^ 1], g->residue, g->reserve, g->flags, g->armed, g->ctr,
                                  g->blk_pack + part_base, g->blk_dead + part_base, g->blk_ndead + part_base, dead_slot,
                                  extra, a, dl.state_in, dl.state0, b == nb - 1 ? 1 : 0)));
      }
      PPRHIP_CHECK_HIP(hipGetLastError());
      part_base += grid;
    }
  }
  k_dense_reduce<<<dim3(1), dim3(1024), 0, g->stream>>>(g->blk_pack, g->blk_dead, g->blk_ndead, part_base, g->ctr,
                                                        out_slot, dead_slot ^ 1, dl.state_in, dl.state0, dl.hist_out,
                                                        dl.state_out, dl.dense_thresh, dl.gs_thresh, pn ? g->pn_ctr : nullptr);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// LDS table of the batched edge kernel.  PPRHIP_SWEEP_HOT_KB (measurement switch): its size in KB, at most 128.
static uint32_t sweep_hot_bytes() {
  static const uint32_t v = [] {
    const char* e = hook_env("PPRHIP_SWEEP_HOT_KB");
    const long kb = e ? atol(e) : 0;
    return kb >= 0 && e && kb * 1024 <= kHotBytes ? (uint32_t)(kb * 1024) : (uint32_t)kHotDefaultBytes;
  }();
  return v;
}

template <int G>
static int launch_dense_edges_bG(pprhip_graph* g, const int32_t* ci, const uint8_t* start_flags,
                                 const uint32_t* chunk_starts, const double* cB, double* accB, const GsBlock& B) {
  if (!g->n_chunks || B.e_hi <= B.e_lo) return PPRHIP_OK;
  const uint32_t hot_max = sweep_hot_bytes() / (8 * G);
  const uint32_t n_hot = g->relabeled ? std::min<uint32_t>(g->n, hot_max) : 0u;
  const uint32_t c_lo = (uint32_t)(B.e_lo / kChunkEdges);
  const uint32_t c_hi = (uint32_t)((B.e_hi + kChunkEdges - 1) / kChunkEdges);
  const uint32_t want = (c_hi - c_lo + 15) / 16;
  const unsigned long long* flags64 = reinterpret_cast<const unsigned long long*>(start_flags);
  if (n_hot) {
    const uint32_t grid = std::min<uint32_t>(want, (uint32_t)g->n_cus);
    k_dense_edges_b<true, G><<<dim3(grid), dim3(1024), sizeof(double) * n_hot * G, g->stream>>>(
        ci, flags64, chunk_starts, c_hi, (unsigned long long)g->m, cB, accB, n_hot, c_lo, B.e_lo, B.e_hi, g->n);
  } else {
    const uint32_t grid = std::min<uint32_t>(want, (uint32_t)g->n_cus * 2u);
    k_dense_edges_b<false, G><<<dim3(grid), dim3(1024), 0, g->stream>>>(
        ci, flags64, chunk_starts, c_hi, (unsigned long long)g->m, cB, accB, 0u, c_lo, B.e_lo, B.e_hi, g->n);
  }
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

#ifdef PPRHIP_TEST_HOOKS
// measurement (PPRHIP_COUNT_LIVE): how many of a sweep's gathers fetch a line that is zero in every column?
// out[0] += out-degrees of the nodes whose line holds a non-zero, out[1] += such nodes
__global__ __launch_bounds__(256) void k_count_live_lines(const double* __restrict__ c8, const uint32_t* __restrict__ out_rp,
                                                          uint32_t n, unsigned long long* out) {
  __shared__ unsigned long long s_red[4];
  const uint32_t v = blockIdx.x * 256u + threadIdx.x;
  unsigned long long d = 0, c = 0;
  if (v < n) {
    bool live = false;
    for (int s = 0; s < kBatch; ++s) live |= c8[(size_t)v * kBatch + s] != 0.0;
    if (live) {
      d = out_rp[v + 1] - out_rp[v];
      c = 1;
    }
  }
  const unsigned long long ds = block_sum_u64(d, s_red), cs = block_sum_u64(c, s_red);
  if (threadIdx.x == 0 && (ds | cs)) {
    atomicAdd(&out[0], ds);
    atomicAdd(&out[1], cs);
  }
}
int launch_count_live_lines(pprhip_graph* P, unsigned long long* d_out) {
  k_count_live_lines<<<dim3((P->n + 255) / 256), dim3(256), 0, P->stream>>>(P->c8[P->c8cur], P->out_rp, P->n, d_out);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}
#endif

#ifdef PPRHIP_TEST_HOOKS
// the edge kernel of one block of a batched forward sweep alone (pprhip_hook_time_sweep_edges)
int launch_sweep_edges_only(pprhip_graph* P, const GsBlock& B) {
  return launch_dense_edges_bG<kBatch>(P, P->in_ci, P->start_flags, P->chunk_starts, P->c8[P->c8cur], P->acc8, B);
}
#endif

int launch_dense_level_b8(pprhip_graph* P, bool backward, const GsBlock* gs_blocks, int n_gs_blocks) {
  PPRHIP_CHECK_HIP(hipMemcpyAsync(P->d_slot_args, P->h_slot_args, sizeof(SlotArgs) * kBatch, hipMemcpyHostToDevice,
                                  P->stream));
  // forward levels pull over the in-CSR, backward levels over the out-CSR
  const int32_t* ci = backward ? P->out_ci : P->in_ci;
  const uint8_t* flags = backward ? P->start_flags_o : P->start_flags;
  const uint32_t* cstarts = backward ? P->chunk_starts_o : P->chunk_starts;
  const int32_t* nz = backward ? P->nz_rows_o : P->nz_rows;
  const int32_t* zr = backward ? P->z_rows_o : P->zin_rows;
  const uint32_t n_nz = backward ? P->n_nz_o : P->n_nz, n_z = backward ? P->n_z_o : P->n_zin;
  const unsigned long long* cross = backward ? P->cross_bits_o : P->cross_bits;
  // slots whose sweep state writes the current contribution array in place; without any, one launch serves the
  // whole sweep (Jacobi rows do not care in which order the blocks run)
  uint32_t gs_mask = 0, entry_mask = 0;
  for (int s = 0; s < kBatch; ++s)
    if (P->h_slot_args[s].active) {
      const int st = P->h_slot_args[s].gs_state;
      if (st == kGsEntry || st == kGsInPlace || st == kGsFlush) gs_mask |= 1u << s;
      if (st == kGsEntry) entry_mask |= 1u << s;
    }
  const uint32_t n_rows = n_nz + n_z;
  const uint32_t n_tiles = (n_rows + kApplyRows - 1) / kApplyRows;
  const GsBlock whole{0u, n_nz, 0ull, (unsigned long long)P->m};
  const bool cut = gs_mask && gs_blocks && n_gs_blocks > 1 && !backward;
  const GsBlock* blocks = cut ? gs_blocks : &whole;
  const int nb = cut ? n_gs_blocks : 1;
  uint32_t part_base = 0;
  for (int b = 0; b < nb; ++b) {
    const GsBlock& B = blocks[b];
    PPRHIP_TRY(launch_dense_edges_bG<kBatch>(P, ci, flags, cstarts, P->c8[P->c8cur], P->acc8, B));
    // block boundaries are multiples of 256 row ordinals, so tiles never straddle; the rows without in-edges
    // follow the last block.  The last block's rows are read by nobody again in this sweep (the next sweep reads the
    // other array), so only the blocks before it write the current array in place.
    const uint32_t t_lo = B.j_lo / kApplyRows;
    const uint32_t t_hi = (b == nb - 1) ? n_tiles : B.j_hi / kApplyRows;
    if (t_hi <= t_lo) continue;
    const uint32_t quota = kApplyBlocks8 / (uint32_t)nb;
    const uint32_t grid = std::max(1u, std::min((t_hi - t_lo + kApplyGroups - 1) / kApplyGroups, quota));
    k_dense_apply_batch<<<dim3(grid), dim3(kApplyThreads), 0, P->stream>>>(
          nz, n_nz, zr, n_z, P->acc8, P->out_rp, backward ? P->in_rp : nullptr, P->c8[P->c8cur], P->c8[P->c8cur ^ 1], t_lo,
          t_hi, b == nb - 1 ? 0u : gs_mask, b == nb - 1 ? 0u : entry_mask, P->d_slot_args, cross, P->prep_bits,
          P->blk_pack8, P->blk_dead8, P->blk_ndead8, part_base, kApplyBlocks8);
    PPRHIP_CHECK_HIP(hipGetLastError());
    part_base += grid;
  }
  k_dense_reduce_batch<<<dim3(kBatch), dim3(1024), 0, P->stream>>>(P->blk_pack8, P->blk_dead8, P->blk_ndead8, part_base,
                                                                   kApplyBlocks8, P->d_slot_args, P->sweep_out);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_compact_prepared(pprhip_graph* g, int cbuf, int out_fbuf, unsigned long long* d_counter, bool backward) {
  const uint32_t grid = grid_for(g->n, 1024, 1024);
  if (g->parent) {
    pprhip_graph* P = g->parent;
    // the rows the batched sweep carries (launch_dense_level_b8): one bit each, per slot
    const uint32_t n_rows = backward ? P->n_nz_o + P->n_z_o : P->n_nz + P->n_zin;
    const uint32_t n_tiles = (n_rows + kApplyRows - 1) / kApplyRows;
    k_compact_bits<<<dim3(grid), dim3(256), 0, g->stream>>>(
        n_rows, backward ? P->n_nz_o : P->n_nz, P->prep_bits + (size_t)g->slot_index * n_tiles,
        backward ? P->nz_rows_o : P->nz_rows, backward ? P->z_rows_o : P->zin_rows, cview(g, cbuf),
        backward ? g->in_rp : g->out_rp, g->F[out_fbuf], g->eoff[out_fbuf], g->cF, d_counter);
  } else {
    k_compact_prepared<<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), cview(g, cbuf), false,
                                                                backward ? g->in_rp : g->out_rp, g->F[out_fbuf],
                                                                g->eoff[out_fbuf], g->cF, d_counter);
  }
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// block partial counts -> ctr->packed[out_slot] (reuses the dense reducer with no dead mass)
static int reduce_partials(pprhip_graph* g, uint32_t n_blocks, int out_slot, int dead_slot, bool with_dead) {
  k_dense_reduce<<<dim3(1), dim3(1024), 0, g->stream>>>(g->blk_pack, with_dead ? g->blk_dead : nullptr, g->blk_ndead,
                                                        n_blocks, g->ctr, out_slot, dead_slot, nullptr, kGsJacobi, nullptr,
                                                        nullptr, 0ull, ~0ull, nullptr);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_count_active(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_slot) {
  const uint32_t grid = grid_for(act_n(g), 256 * 8, 1024);
  if (seed_kind == 0)
    k_count_active<0><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->out_rp, g->flags, g->armed,
                                                               g->blk_pack, &g->ctr->hist[kMaxBatch + 2], a);
  else
    k_count_active<1><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->out_rp, g->flags, g->armed,
                                                               g->blk_pack, &g->ctr->hist[kMaxBatch + 2], a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return reduce_partials(g, grid, out_slot, 0, false);
}

bool old_small_kernels() {  // PPRHIP_TOPK_OLD_PASSES=1 (measurement switch): the two-pass kernels of rounds 1-4
  static const bool v = hook_env("PPRHIP_TOPK_OLD_PASSES") != nullptr;
  return v;
}

int launch_seed_list(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_fbuf, unsigned long long* d_counter,
                     bool write_armed) {
  // 1024 nodes per workgroup (fewer or more were slower); a slot of a threaded batch keeps the cap of 1024
  // workgroups: with sixteen queries on the chip, smaller launches do better (1 018 vs 946-977 queries/s)
  const uint32_t grid = grid_for(act_n(g), 1024, g->sync ? 1024 : 16384);
  if (seed_kind == 0)
    k_seed_list<0><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->out_rp, g->flags, g->F[out_fbuf],
                                                            g->eoff[out_fbuf], d_counter, nullptr, a);
  else if (write_armed && !old_small_kernels()) {
    // (the one-pass form always rewrites the flags and the armed bits of every node: the round-start call)
    const uint32_t tiles = (act_n(g) + 256u * kSeedItems - 1) / (256u * kSeedItems);
    k_seed_list_topk<<<dim3(std::max(1u, tiles)), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->out_rp, g->flags,
                                                                            g->F[out_fbuf], g->eoff[out_fbuf], d_counter,
                                                                            g->armed, a);
  } else
    k_seed_list<1><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->out_rp, g->flags, g->F[out_fbuf],
                                                            g->eoff[out_fbuf], d_counter, write_armed ? g->armed : nullptr, a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_seed_dense(pprhip_graph* g, const PushArgs& a, int seed_kind, int cbuf, int out_slot, int dead_slot) {
  const uint32_t grid = grid_for(act_n(g), 256 * 8, 1024);
  if (seed_kind == 0)
    k_seed_dense<0><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->reserve, g->out_rp, g->flags,
                                                             cview(g, cbuf), g->blk_pack, g->blk_dead,
                                                             g->blk_ndead, a);
  else
    k_seed_dense<1><<<dim3(grid), dim3(256), 0, g->stream>>>(act_n(g), g->residue, g->reserve, g->out_rp, g->flags,
                                                             cview(g, cbuf), g->blk_pack, g->blk_dead,
                                                             g->blk_ndead, a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return reduce_partials(g, grid, out_slot, dead_slot, true);
}

int launch_sum(pprhip_graph* g, const double* x, uint32_t n) {
  const uint32_t np = grid_for(n, 256 * 16, 1024);
  k_sum_partial<<<dim3(np), dim3(256), 0, g->stream>>>(x, n, g->partial);
  k_sum_final<<<dim3(1), dim3(256), 0, g->stream>>>(g->partial, np, g->ctr);
  PPRHIP_CHECK_HIP(hipGetLastError());
  g->sum_np = 0;
  return PPRHIP_OK;
}

// the partial sums only: the walk plan that follows (launch_mc_plan with the budget derived on the device) adds them up
// itself, in every workgroup - one launch less on the chain of a top-k round
int launch_sum_partial(pprhip_graph* g, const double* x, uint32_t n) {
  if (old_small_kernels()) return launch_sum(g, x, n);
  const uint32_t np = grid_for(n, 256 * 8, 1024);
  k_sum_partial<<<dim3(np), dim3(256), 0, g->stream>>>(x, n, g->partial);
  PPRHIP_CHECK_HIP(hipGetLastError());
  g->sum_np = np;
  return PPRHIP_OK;
}

int launch_permute_out(pprhip_graph* g, const double* x, double* out) {
  k_permute_out<<<dim3(grid_for(g->n, 256, 4096)), dim3(256), 0, g->stream>>>(x, g->old2new, out, g->n);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// Current device: code object loaded, large dynamic LDS opted in (above 64 KB it needs an explicit opt-in per
// device).  Called once per device under the graph-lift lock (engine.cpp), never from a launch path.
int init_kernels_push() {
  PPRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_edges<true, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kHotMax)));
  PPRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_edges<true, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * kHotMax)));
  PPRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_edges_b<true, kBatch>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kHotBytes));
  PPRHIP_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense_edges_panel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kPanelLdsBytes));
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_sparse_push<kBackward>)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_dense_apply_batch)));
  return PPRHIP_OK;
}

int launch_set_f64(pprhip_graph* g, double* p, uint32_t idx, double value) {
  k_set_f64<<<dim3(1), dim3(1), 0, g->stream>>>(p, idx, value);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
