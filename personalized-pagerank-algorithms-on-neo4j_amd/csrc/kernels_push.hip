// kernels_push.hip — frontier push kernels for gfx950 (MI355X).
//
// One *level* pushes every frontier node at once from its residue at level start (the
// frontier-synchronous form of Forward_Push.java:86-139 / Backward_Search.java:58-96).  A level
// runs in one of two shapes:
//
//   sparse  k_sparse_prepare (per frontier node: take the residue, credit the reserve, compute the
//           per-edge contribution) then k_sparse_push (edge-parallel over the frontier's edges:
//           coalesced col_idx reads, one returning fp64 atomic per edge, threshold-crossing
//           detection on (old, old + c), wave-aggregated append to the next frontier);
//   dense   k_hub_pull + k_dense_tiles: a pull sweep over the in-CSR in row-aligned tiles of
//           <= 2048 edges; contributions are gathered through LDS and every row is applied and, if
//           it crosses the threshold, prepared for the next level in the same kernel (no atomics
//           on the residue vector).
//
// HBM-bound integer/fp64 work: no MFMA anywhere.  All arithmetic is IEEE double with
// -ffp-contract=off so each product / quotient rounds exactly as the reference's Java does.
#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

constexpr int kPushTile = 2048;  // edges per workgroup iteration of k_sparse_push
constexpr int kStageCap = 512;   // frontier entries staged in LDS at a time

// ------------------------------------------------------------------------------------------------
// sparse level, step 1: every frontier node gives up its residue
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_prepare(const int32_t* __restrict__ F, uint32_t nf,
                                                         const uint32_t* __restrict__ out_rp,
                                                         double* __restrict__ res, double* __restrict__ reserve,
                                                         double* __restrict__ cF, double* __restrict__ c_dense,
                                                         DevCounters* ctr, int dead_slot, PushArgs a) {
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  double dead = 0.0;
  unsigned long long ndead = 0;
  if (i < nf) {
    const int32_t v = F[i];
    const double rc = res[v];
    res[v] = 0.0;                         // Forward_Push.java:89
    reserve[v] = reserve[v] + rc * a.alpha;  // :91-95
    double c;
    if (MODE == kBackward) {
      c = (1.0 - a.alpha) * rc;  // Backward_Search.java:72 (divided by d_out(u) per edge)
    } else {
      const uint32_t d = out_rp[v + 1] - out_rp[v];
      if (d == 0) {  // Forward_Push.java:101-104: the mass goes back to the source
        c = 0.0;
        dead = rc * (1.0 - a.alpha);
        ndead = 1;
      } else {
        c = ((1.0 - a.alpha) * rc) / (double)d;  // :117
      }
    }
    if (c_dense)
      c_dense[v] = c;
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

// ------------------------------------------------------------------------------------------------
// sparse level, step 2: contributions land edge by edge
// ------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void push_one(bool valid, int32_t u, double c, const uint32_t* __restrict__ out_rp,
                                         const uint32_t* __restrict__ in_rp, double* __restrict__ res,
                                         uint8_t* __restrict__ flags, int32_t* __restrict__ Fn,
                                         uint32_t* __restrict__ eoffn, unsigned long long* out_counter,
                                         const PushArgs& a) {
  bool crossing = false;
  uint32_t adeg = 0;
  if (valid) {
    const uint32_t du = out_rp[u + 1] - out_rp[u];
    if (MODE == kBackward) {
      const double add = c / (double)du;  // Backward_Search.java:84-85
      const double old = atomic_add_ret(&res[u], add);
      const double nw = old + add;
      crossing = !(old > a.rmax) && (nw > a.rmax);  // :89 strict, un-normalised
      if (crossing) adeg = in_rp[u + 1] - in_rp[u];
    } else {
      const double old = atomic_add_ret(&res[u], c);  // Forward_Push.java:123-127
      const double nw = old + c;
      crossing = !active_fwd(old, du, a.rmax) && active_fwd(nw, du, a.rmax);  // :132
      adeg = du;
      if (MODE == kFwdTopk && active_fwd(nw, du, a.min_rmax)) flags[u] = 1;  // :232-237 (parked)
    }
  }
  wave_append(crossing, u, adeg, Fn, eoffn, out_counter);
}

template <int MODE>
__global__ __launch_bounds__(256) void k_sparse_push(const int32_t* __restrict__ F, const double* __restrict__ cF,
                                                      const uint32_t* __restrict__ eoff,
                                                      const unsigned long long* __restrict__ in_counter,
                                                      const uint32_t* __restrict__ trp, const int32_t* __restrict__ tci,
                                                      const uint32_t* __restrict__ out_rp,
                                                      const uint32_t* __restrict__ in_rp, double* __restrict__ res,
                                                      uint8_t* __restrict__ flags, int32_t* __restrict__ Fn,
                                                      uint32_t* __restrict__ eoffn, DevCounters* ctr, int out_slot,
                                                      int dead_slot, PushArgs a) {
  __shared__ uint32_t s_eoff[kStageCap + 1];
  __shared__ uint32_t s_row[kStageCap];
  __shared__ double s_c[kStageCap];
  __shared__ uint32_t s_i0;
  const int tid = threadIdx.x;
  const unsigned long long pk = *in_counter;
  const uint32_t nf = (uint32_t)(pk >> kPackShift);
  const unsigned long long E = pk & kPackMask;
  unsigned long long* out_counter = &ctr->packed[out_slot];

  if (MODE != kBackward && blockIdx.x == 0 && wave_id() == 0) {
    // dead-end mass of this level lands on the source (Forward_Push.java:101-113)
    const double dead = ctr->dead[dead_slot];
    const bool valid = (lane_id() == 0) && (dead > 0.0);
    push_one<MODE>(valid, a.src, dead, out_rp, in_rp, res, flags, Fn, eoffn, out_counter, a);
    if (valid) ctr->dead[dead_slot] = 0.0;
  }

  const unsigned long long n_tiles = (E + kPushTile - 1) / kPushTile;
  for (unsigned long long t = blockIdx.x; t < n_tiles; t += gridDim.x) {
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
      const uint32_t span = cov_hi > ce ? (uint32_t)(cov_hi - ce) : 0u;
      const uint32_t iters = (span + 255u) >> 8;
      for (uint32_t it = 0; it < iters; ++it) {
        const unsigned long long e = ce + (unsigned long long)it * 256ull + tid;
        const bool valid = e < cov_hi;
        int32_t u = 0;
        double c = 0.0;
        if (valid) {
          const uint32_t e32 = (uint32_t)e;
          uint32_t lo = 0, hi = cnt;
          while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (s_eoff[mid] <= e32) lo = mid + 1; else hi = mid;
          }
          const uint32_t j = lo - 1;
          u = tci[s_row[j] + (e32 - s_eoff[j])];
          c = s_c[j];
        }
        push_one<MODE>(valid, u, c, out_rp, in_rp, res, flags, Fn, eoffn, out_counter, a);
      }
      __syncthreads();
      ce = cov_hi;
      ci0 += cnt;
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// dense level: pull sweep over the in-CSR
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hub_pull(const uint32_t* __restrict__ chunks, uint32_t n_chunks,
                                                   const int32_t* __restrict__ in_ci,
                                                   const double* __restrict__ c_cur, double* __restrict__ hubacc) {
  __shared__ double s_red[4];
  for (uint32_t ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
    const uint32_t h = chunks[3 * ch], e0 = chunks[3 * ch + 1], e1 = chunks[3 * ch + 2];
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    uint32_t e = e0 + threadIdx.x;
    for (; e + 768 < e1; e += 1024) {
      const int32_t i0 = in_ci[e], i1 = in_ci[e + 256], i2 = in_ci[e + 512], i3 = in_ci[e + 768];
      acc0 += c_cur[i0];
      acc1 += c_cur[i1];
      acc2 += c_cur[i2];
      acc3 += c_cur[i3];
    }
    for (; e < e1; e += 256) acc0 += c_cur[in_ci[e]];
    const double s = block_sum_f64((acc0 + acc1) + (acc2 + acc3), s_red);
    if (threadIdx.x == 0 && s != 0.0) atomic_add_noret(&hubacc[h], s);
  }
}

__device__ __forceinline__ int swz(int k) { return k ^ ((k >> 5) & 31); }

template <int MODE>
__global__ __launch_bounds__(256) void k_dense_tiles(const uint32_t* __restrict__ tile_row, uint32_t n_tiles,
                                                      const uint32_t* __restrict__ in_rp,
                                                      const int32_t* __restrict__ in_ci,
                                                      const uint32_t* __restrict__ out_rp,
                                                      const double* __restrict__ c_cur, double* __restrict__ c_next,
                                                      double* __restrict__ res, double* __restrict__ reserve,
                                                      uint8_t* __restrict__ flags, const int32_t* __restrict__ hub_rows,
                                                      uint32_t n_hubs, double* __restrict__ hubacc, DevCounters* ctr,
                                                      int out_slot, int dead_slot, PushArgs a) {
  __shared__ double s_val[kTileEdges];
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  const int tid = threadIdx.x;
  const uint32_t t = blockIdx.x;
  bool have = false;
  int32_t u = -1;
  double acc = 0.0;
  if (t < n_tiles) {
    const uint32_t r0 = tile_row[t], r1 = tile_row[t + 1];
    const uint32_t e0 = in_rp[r0], e1 = in_rp[r1];
    const uint32_t ne = e1 - e0;
    if (ne <= (uint32_t)kTileEdges) {  // a hub row is a tile of its own and is applied by the hub blocks
      int32_t idx[8];
      double val[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t k = tid + 256 * j;
        idx[j] = k < ne ? in_ci[e0 + k] : -1;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) val[j] = idx[j] >= 0 ? c_cur[idx[j]] : 0.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t k = tid + 256 * j;
        if (k < ne) s_val[swz((int)k)] = val[j];
      }
      __syncthreads();
      if ((uint32_t)tid < r1 - r0) {
        u = (int32_t)(r0 + tid);
        const int kb = (int)(in_rp[u] - e0), ke = (int)(in_rp[u + 1] - e0);
        for (int k = kb; k < ke; ++k) acc += s_val[swz(k)];
        have = true;
      }
    }
  } else {
    const uint32_t h = (t - n_tiles) * 256u + tid;
    if (h < n_hubs) {
      u = hub_rows[h];
      acc = hubacc[h];
      hubacc[h] = 0.0;
      have = true;
    }
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
    if (acc > 0.0) {
      const double old = res[u];
      const double nw = old + acc;
      const bool crossing = (MODE == kPower) ? true : (!active_fwd(old, d, a.rmax) && active_fwd(nw, d, a.rmax));
      if (MODE == kFwdTopk && active_fwd(nw, d, a.min_rmax)) flags[u] = 1;
      if (crossing) {  // becomes a frontier node of the next level: prepare it right here
        reserve[u] = reserve[u] + nw * a.alpha;
        res[u] = 0.0;
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
  }
  const double ds = block_sum_f64(dead_next, s_red);
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  if (tid == 0) {
    if (ps) atomic_add_u64(&ctr->packed[out_slot], ps);
    if (nd) {
      atomic_add_noret(&ctr->dead[dead_slot ^ 1], ds);
      atomic_add_u64(&ctr->dead_pops, nd);
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
                                          uint8_t* __restrict__ flags, const PushArgs& a, bool mutate) {
  const double r = res[v];
  if (KIND == 0) return active_fwd(r, d, a.rmax);
  if (!flags[v]) return false;
  if (active_fwd(r, d, a.rmax)) {
    if (mutate) flags[v] = 0;
    return true;
  }
  if (mutate && !active_fwd(r, d, a.min_rmax)) flags[v] = 0;
  return false;
}

template <int KIND>
__global__ __launch_bounds__(256) void k_count_active(uint32_t n, const double* __restrict__ res,
                                                       const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                       DevCounters* ctr, int out_slot, PushArgs a) {
  __shared__ unsigned long long s_red2[4];
  unsigned long long pack = 0;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t d = out_rp[v + 1] - out_rp[v];
    if (seed_pred<KIND>(v, res, d, flags, a, false)) pack += (1ull << kPackShift) | (unsigned long long)d;
  }
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  if (threadIdx.x == 0 && ps) atomic_add_u64(&ctr->packed[out_slot], ps);
}

template <int KIND>
__global__ __launch_bounds__(256) void k_seed_list(uint32_t n, const double* __restrict__ res,
                                                    const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                    int32_t* __restrict__ Fn, uint32_t* __restrict__ eoffn,
                                                    DevCounters* ctr, int out_slot, PushArgs a) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nround = (n + stride - 1) / stride * stride;  // keep whole waves convergent
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nround; v += stride) {
    bool take = false;
    uint32_t d = 0;
    if (v < n) {
      d = out_rp[v + 1] - out_rp[v];
      take = seed_pred<KIND>(v, res, d, flags, a, true);
    }
    wave_append(take, (int32_t)v, d, Fn, eoffn, &ctr->packed[out_slot]);
  }
}

template <int KIND>
__global__ __launch_bounds__(256) void k_seed_dense(uint32_t n, double* __restrict__ res, double* __restrict__ reserve,
                                                     const uint32_t* __restrict__ out_rp, uint8_t* __restrict__ flags,
                                                     double* __restrict__ c_dense, DevCounters* ctr, int out_slot,
                                                     int dead_slot, PushArgs a) {
  __shared__ double s_red[4];
  __shared__ unsigned long long s_red2[4];
  double dead = 0.0;
  unsigned long long pack = 0, ndead = 0;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < n; v += gridDim.x * blockDim.x) {
    const uint32_t d = out_rp[v + 1] - out_rp[v];
    double c = 0.0;
    if (seed_pred<KIND>(v, res, d, flags, a, true)) {
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
    c_dense[v] = c;
  }
  const double ds = block_sum_f64(dead, s_red);
  const unsigned long long ps = block_sum_u64(pack, s_red2);
  const unsigned long long nd = block_sum_u64(ndead, s_red2);
  if (threadIdx.x == 0) {
    if (ps) atomic_add_u64(&ctr->packed[out_slot], ps);
    if (nd) {
      atomic_add_noret(&ctr->dead[dead_slot], ds);
      atomic_add_u64(&ctr->dead_pops, nd);
    }
  }
}

// dense-prepared state -> sparse-prepared state: list every node holding a contribution
__global__ __launch_bounds__(256) void k_compact_prepared(uint32_t n, const double* __restrict__ c_dense,
                                                           const uint32_t* __restrict__ trp, int32_t* __restrict__ Fn,
                                                           uint32_t* __restrict__ eoffn, double* __restrict__ cF,
                                                           DevCounters* ctr, int out_slot) {
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nround = (n + stride - 1) / stride * stride;
  for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < nround; v += stride) {
    double c = 0.0;
    uint32_t d = 0;
    if (v < n) {
      c = c_dense[v];
      if (c > 0.0) d = trp[v + 1] - trp[v];
    }
    const uint32_t pos = wave_append(c > 0.0, (int32_t)v, d, Fn, eoffn, &ctr->packed[out_slot]);
    if (c > 0.0) cF[pos] = c;
  }
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

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline uint32_t grid_for(uint64_t work, uint32_t per_block, uint32_t cap) {
  uint64_t b = (work + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (uint32_t)b;
}

#define DISPATCH_MODE(MODEVAR, ...)                                       \
  switch (MODEVAR) {                                                      \
    case kFwdWhole: { constexpr int M = kFwdWhole; __VA_ARGS__; } break;  \
    case kFwdTopk: { constexpr int M = kFwdTopk; __VA_ARGS__; } break;    \
    case kBackward: { constexpr int M = kBackward; __VA_ARGS__; } break;  \
    default: { constexpr int M = kPower; __VA_ARGS__; } break;            \
  }

int launch_sparse_prepare(pprhip_graph* g, const PushArgs& a, int fbuf, uint32_t nf, bool scatter_dense, int cbuf,
                          int dead_slot) {
  if (nf == 0) return PPRHIP_OK;
  const uint32_t grid = (nf + 255) / 256;
  double* cd = scatter_dense ? g->cdense[cbuf] : nullptr;
  DISPATCH_MODE(a.mode, k_sparse_prepare<M><<<dim3(grid), dim3(256), 0, g->stream>>>(
                            g->F[fbuf], nf, g->out_rp, g->residue, g->reserve, g->cF, cd, g->ctr, dead_slot, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_sparse_push(pprhip_graph* g, const PushArgs& a, int fbuf, const unsigned long long* d_in_counter,
                       uint64_t ef_upper, int out_fbuf, int out_slot, int dead_slot) {
  const uint32_t grid = grid_for(ef_upper, kPushTile, 2048);
  const bool bwd = a.mode == kBackward;
  const uint32_t* trp = bwd ? g->in_rp : g->out_rp;
  const int32_t* tci = bwd ? g->in_ci : g->out_ci;
  DISPATCH_MODE(a.mode, k_sparse_push<M><<<dim3(grid), dim3(256), 0, g->stream>>>(
                            g->F[fbuf], g->cF, g->eoff[fbuf], d_in_counter, trp, tci, g->out_rp, g->in_rp, g->residue,
                            g->flags, g->F[out_fbuf], g->eoff[out_fbuf], g->ctr, out_slot, dead_slot, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_dense_level(pprhip_graph* g, const PushArgs& a, int cbuf, int out_slot, int dead_slot) {
  if (g->n_hub_chunks) {
    hipLaunchKernelGGL(k_hub_pull, dim3(grid_for(g->n_hub_chunks, 1, 4096)), dim3(256), 0, g->stream, g->hub_chunks,
                       g->n_hub_chunks, g->in_ci, g->cdense[cbuf], g->hubacc);
    PPRHIP_CHECK_HIP(hipGetLastError());
  }
  const uint32_t grid = g->n_tiles + (g->n_hubs + 255) / 256;
  DISPATCH_MODE(a.mode, k_dense_tiles<M><<<dim3(grid), dim3(256), 0, g->stream>>>(
                            g->tile_row, g->n_tiles, g->in_rp, g->in_ci, g->out_rp, g->cdense[cbuf],
                            g->cdense[cbuf ^ 1], g->residue, g->reserve, g->flags, g->hub_rows, g->n_hubs, g->hubacc,
                            g->ctr, out_slot, dead_slot, a));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_compact_prepared(pprhip_graph* g, int cbuf, int out_fbuf, int out_slot, bool backward) {
  const uint32_t grid = grid_for(g->n, 256, 2048);
  hipLaunchKernelGGL(k_compact_prepared, dim3(grid), dim3(256), 0, g->stream, g->n, g->cdense[cbuf],
                     backward ? g->in_rp : g->out_rp, g->F[out_fbuf], g->eoff[out_fbuf], g->cF, g->ctr, out_slot);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_count_active(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_slot) {
  const uint32_t grid = grid_for(g->n, 256, 2048);
  if (seed_kind == 0)
    hipLaunchKernelGGL(k_count_active<0>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->out_rp, g->flags,
                       g->ctr, out_slot, a);
  else
    hipLaunchKernelGGL(k_count_active<1>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->out_rp, g->flags,
                       g->ctr, out_slot, a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_seed_list(pprhip_graph* g, const PushArgs& a, int seed_kind, int out_fbuf, int out_slot) {
  const uint32_t grid = grid_for(g->n, 256, 2048);
  if (seed_kind == 0)
    hipLaunchKernelGGL(k_seed_list<0>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->out_rp, g->flags,
                       g->F[out_fbuf], g->eoff[out_fbuf], g->ctr, out_slot, a);
  else
    hipLaunchKernelGGL(k_seed_list<1>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->out_rp, g->flags,
                       g->F[out_fbuf], g->eoff[out_fbuf], g->ctr, out_slot, a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_seed_dense(pprhip_graph* g, const PushArgs& a, int seed_kind, int cbuf, int out_slot, int dead_slot) {
  const uint32_t grid = grid_for(g->n, 256, 2048);
  if (seed_kind == 0)
    hipLaunchKernelGGL(k_seed_dense<0>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->reserve, g->out_rp,
                       g->flags, g->cdense[cbuf], g->ctr, out_slot, dead_slot, a);
  else
    hipLaunchKernelGGL(k_seed_dense<1>, dim3(grid), dim3(256), 0, g->stream, g->n, g->residue, g->reserve, g->out_rp,
                       g->flags, g->cdense[cbuf], g->ctr, out_slot, dead_slot, a);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_sum(pprhip_graph* g, const double* x, uint32_t n) {
  const uint32_t np = grid_for(n, 256 * 16, 1024);
  hipLaunchKernelGGL(k_sum_partial, dim3(np), dim3(256), 0, g->stream, x, n, g->partial);
  hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(256), 0, g->stream, g->partial, np, g->ctr);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_set_f64(pprhip_graph* g, double* p, uint32_t idx, double value) {
  hipLaunchKernelGGL(k_set_f64, dim3(1), dim3(1), 0, g->stream, p, idx, value);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
