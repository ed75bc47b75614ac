// device_utils.hpp — wave64 / workgroup primitives shared by the pprhip kernels (gfx950).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

namespace pprhip {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// The reference's enqueue test `r / (double)d >= rmax` (Forward_Push.java:109,132); d == 0 makes
// the quotient +Inf for r > 0 and NaN for r == 0.
__device__ __forceinline__ bool active_fwd(double r, uint32_t d, double rmax) {
  return d > 0 ? (r / (double)d >= rmax) : (r > 0.0);
}

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = __shfl_up(x, o);
    if (lane >= o) x += t;
  }
  return x;
}

__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long x) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long t = __shfl_up(x, o);
    if (lane >= o) x += t;
  }
  return x;
}

__device__ __forceinline__ double wave_sum_f64(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  return x;  // valid in lane 0
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  return x;
}

// Workgroup sums; result valid in thread 0.  `scratch` holds one slot per wave.
__device__ __forceinline__ double block_sum_f64(double x, double* scratch) {
  x = wave_sum_f64(x);
  const int nw = (blockDim.x + 63) >> 6;
  if (lane_id() == 0) scratch[wave_id()] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) s += scratch[w];
  __syncthreads();
  return s;
}

__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long x, unsigned long long* scratch) {
  x = wave_sum_u64(x);
  const int nw = (blockDim.x + 63) >> 6;
  if (lane_id() == 0) scratch[wave_id()] = x;
  __syncthreads();
  unsigned long long s = 0;
  if (threadIdx.x == 0)
    for (int w = 0; w < nw; ++w) s += scratch[w];
  __syncthreads();
  return s;
}

// fp64 atomics at agent scope.  -munsafe-fp-atomics makes both lower to global_atomic_add_f64.
__device__ __forceinline__ double atomic_add_ret(double* p, double v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_noret(double* p, double v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long atomic_add_u64(unsigned long long* p, unsigned long long v) {
  return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}


// ---- DPP cross-lane helpers (gfx9 family: row_shr within 16-lane rows, row_bcast across rows).
// A wave scan built from these costs a few dozen VALU instructions; the generic __shfl_up path
// goes through ds_bpermute and a wait per step.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v));
  const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// inclusive scan of one u32 per lane over the wave
__device__ __forceinline__ uint32_t wave_incl_scan_u32_dpp(uint32_t x) {
  x += (uint32_t)dpp_i32<0x111, 0xf>((int)x);  // row_shr:1
  x += (uint32_t)dpp_i32<0x112, 0xf>((int)x);  // row_shr:2
  x += (uint32_t)dpp_i32<0x114, 0xf>((int)x);  // row_shr:4
  x += (uint32_t)dpp_i32<0x118, 0xf>((int)x);  // row_shr:8
  x += (uint32_t)dpp_i32<0x142, 0xa>((int)x);  // row_bcast:15 -> rows 1, 3
  x += (uint32_t)dpp_i32<0x143, 0xc>((int)x);  // row_bcast:31 -> rows 2, 3
  return x;
}

// inclusive segmented sum over the wave: S(l) = x(l) + (head(l) ? 0 : S(l-1))
__device__ __forceinline__ double wave_seg_scan_f64_dpp(double val, bool head) {
  int flag = head ? 1 : 0;
#define PPRHIP_SEG_STEP(CTRL, RM)                       \
  {                                                     \
    const double tv = dpp_f64<CTRL, RM>(val);           \
    const int tf = dpp_i32<CTRL, RM>(flag);             \
    if (!flag) {                                        \
      val += tv;                                        \
      flag = tf;                                        \
    }                                                   \
  }
  PPRHIP_SEG_STEP(0x111, 0xf)
  PPRHIP_SEG_STEP(0x112, 0xf)
  PPRHIP_SEG_STEP(0x114, 0xf)
  PPRHIP_SEG_STEP(0x118, 0xf)
  PPRHIP_SEG_STEP(0x142, 0xa)
  PPRHIP_SEG_STEP(0x143, 0xc)
#undef PPRHIP_SEG_STEP
  return val;
}

// value of the previous lane (0 for lane 0)
__device__ __forceinline__ double wave_prev_f64_dpp(double v) { return dpp_f64<0x138, 0xf>(v); }  // wave_shr:1

// Wave-aggregated append of (node, degree) pairs to a frontier.  One packed 64-bit atomic per
// wave reserves both the list slots (high bits) and the edge range (low 36 bits), so the list's
// edge offsets are an exclusive prefix sum in list order without any scan kernel.
// Must be called by all 64 lanes of a wave (convergent); `take` selects the appending lanes.
// Returns the list position of the calling lane's entry (0xFFFFFFFF for lanes that append nothing).
__device__ __forceinline__ uint32_t wave_append(bool take, int32_t node, uint32_t deg, int32_t* __restrict__ list,
                                                uint32_t* __restrict__ eoff, unsigned long long* packed) {
  const unsigned long long mask = __ballot(take);
  if (mask == 0) return 0xFFFFFFFFu;
  const int lane = lane_id();
  const uint32_t myd = take ? deg : 0u;
  const uint32_t incl = wave_incl_scan_u32(myd);
  const uint32_t total = __shfl(incl, 63);
  const int leader = __ffsll((long long)mask) - 1;
  unsigned long long base = 0;
  if (lane == leader)
    base = atomic_add_u64(packed, ((unsigned long long)__popcll(mask) << kPackShift) | (unsigned long long)total);
  base = __shfl(base, leader);
  if (take) {
    const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
    const uint32_t pos = (uint32_t)(base >> kPackShift) + rank;
    list[pos] = node;
    eoff[pos] = (uint32_t)(base & kPackMask) + incl - myd;
    return pos;
  }
  return 0xFFFFFFFFu;
}


// Workgroup exclusive scan of one value per thread (256 threads); returns the exclusive prefix and
// writes the workgroup total to *total.  `scratch` holds 4 slots.
template <class T>
__device__ __forceinline__ T block_excl_scan_256(T x, T* scratch, T* total) {
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
  for (int w = 0; w < 4; ++w) {
    const T s = scratch[w];
    if (w < wv) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - x;
}

// Two-pass compaction of the contiguous item range owned by one workgroup (256 threads).
// eval(i, &weight) says whether item i is taken and what it weighs (degree / walk count); pass 1
// totals the range, ONE packed atomic reserves list slots (high bits) and the weight range (low 36
// bits), pass 2 re-evaluates and emits items in ascending order with their exclusive weight
// offsets.  eval must be a pure function of data that does not change during the kernel.
template <class Eval, class Emit>
__device__ __forceinline__ void block_range_compact(uint32_t lo, uint32_t hi, unsigned long long* counter, Eval eval,
                                                    Emit emit) {
  __shared__ unsigned long long s_scan[4];
  __shared__ unsigned long long s_base, s_tot;
  unsigned long long pack = 0;
  for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
    unsigned long long w = 0;
    if (eval(i, &w)) pack += (1ull << kPackShift) | w;
  }
  const unsigned long long tot = block_sum_u64(pack, s_scan);  // valid in thread 0
  if (threadIdx.x == 0) {
    s_tot = tot;
    s_base = tot ? atomic_add_u64(counter, tot) : 0ull;
  }
  __syncthreads();
  if (s_tot == 0) return;
  unsigned long long run = s_base;  // running (count << 36 | weight) offset, uniform across the workgroup
  for (uint32_t c = lo; c < hi; c += 256) {
    const uint32_t i = c + threadIdx.x;
    unsigned long long w = 0;
    const bool take = (i < hi) && eval(i, &w);
    const unsigned long long mine = take ? ((1ull << kPackShift) | w) : 0ull;
    unsigned long long chunk_total = 0;
    const unsigned long long excl = block_excl_scan_256<unsigned long long>(mine, s_scan, &chunk_total);
    if (take) {
      const unsigned long long at = run + excl;
      emit(i, (uint32_t)(at >> kPackShift), at & kPackMask, w);
    }
    run += chunk_total;
  }
}

// One-pass compaction of a TILE of 256 x ITEMS consecutive items by one workgroup (round 5).  Thread t owns the ITEMS
// consecutive items tile_lo + t * ITEMS ...; the caller has evaluated them into registers already (all of a thread's
// loads in flight together, no re-evaluation): take[i] / w[i] say whether item i is taken and what it weighs.  One
// workgroup scan over the threads' packed totals, ONE packed atomic for the tile's list slots and weight range, and
// emit(i, position, weight offset) for the taken items, in ascending item order.  block_range_compact walks its range
// twice with a workgroup scan per 256 items: on the passes of a top-k round (2 M nodes, a few thousand of them taken)
// that is a dozen barriers and dependent load chains per workgroup, 40 us for 26 MB of traffic.
template <int ITEMS, class Emit>
__device__ __forceinline__ void block_tile_compact(const bool (&take)[ITEMS], const unsigned long long (&w)[ITEMS],
                                                   unsigned long long* counter, Emit emit) {
  __shared__ unsigned long long s_scan[4];
  __shared__ unsigned long long s_base;
  unsigned long long mine = 0;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (take[i]) mine += (1ull << kPackShift) | w[i];
  unsigned long long total = 0;
  const unsigned long long excl = block_excl_scan_256<unsigned long long>(mine, s_scan, &total);
  if (total == 0) return;  // (uniform: every thread holds the same total)
  if (threadIdx.x == 0) s_base = atomic_add_u64(counter, total);
  __syncthreads();
  unsigned long long at = s_base + excl;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i)
    if (take[i]) {
      emit(i, (uint32_t)(at >> kPackShift), at & kPackMask);
      at += (1ull << kPackShift) | w[i];
    }
}

}  // namespace pprhip
