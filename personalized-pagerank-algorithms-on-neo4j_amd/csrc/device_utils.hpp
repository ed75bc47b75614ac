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

}  // namespace pprhip
