// l2_policy_rate.hip — developer micro-benchmark (round 5): can cold gathers be kept from evicting hot lines from L2?
// The batched sweep gathers 128-byte lines c8[v][0..15]: 54 % of them from the 32 K hottest vertices (4 MB: one L2's
// worth), the rest from a tail of 2 M lines that are touched once or twice per sweep and still allocate in L2 on
// their way through (TCC hit rate 0.31).  Here every lane group (16 lanes x 8 B = one line) alternates between a hot
// table (4 MB) and a cold table (1 GB), 50 : 50, and the cold gathers are issued as
//   plain    ordinary loads (what the sweep does),
//   nt       non-temporal loads (__builtin_nontemporal_load),
//   uncached ordinary loads from memory allocated with hipDeviceMallocUncached.
// Higher G lines/s = the hot half kept its hits.
//   hipcc --offload-arch=gfx950 -O3 -o l2_policy_rate tools/micro/l2_policy_rate.hip && ./l2_policy_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <bool NT>
__global__ __launch_bounds__(1024) void k_mixed(const double* __restrict__ hot, uint32_t hot_rows, const double* __restrict__ cold,
                                                uint32_t cold_rows, int steps, int cold_of_8, double* out) {
  const uint32_t grp = (blockIdx.x * 1024u + threadIdx.x) / 16u, sub = threadIdx.x % 16u;
  double acc = 0.0;
  uint32_t s = mix(grp + 0x9e3779b9u);
  for (int t = 0; t < steps; ++t) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = mix(s + 0x85ebca6bu * (uint32_t)(i + 1));
      if (i < cold_of_8) {
        const double* p = &cold[(size_t)(h % cold_rows) * 16u + sub];
        v[i] = NT ? __builtin_nontemporal_load(p) : *p;
      } else {
        v[i] = hot[(size_t)(h % hot_rows) * 16u + sub];
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc;
}

template <class F>
static float best_of(F launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(a);
    launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t hot_bytes = (size_t)4 << 20, cold_bytes = (size_t)1 << 30;
  double *hot, *cold, *cold_uc = nullptr, *out;
  if (hipMalloc(&hot, hot_bytes) != hipSuccess || hipMalloc(&cold, cold_bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) return 1;
  const hipError_t e = hipExtMallocWithFlags((void**)&cold_uc, cold_bytes, hipDeviceMallocUncached);
  (void)hipMemset(hot, 0, hot_bytes);
  (void)hipMemset(cold, 0, cold_bytes);
  if (e == hipSuccess) (void)hipMemset(cold_uc, 0, cold_bytes);
  else printf("hipExtMallocWithFlags(hipDeviceMallocUncached) failed: %s\n", hipGetErrorString(e));
  const uint32_t hr = (uint32_t)(hot_bytes / 128), cr = (uint32_t)(cold_bytes / 128);
  const int grid = 256, steps = 64;
  const double lines = (double)grid * 1024 / 16 * steps * 8;
  printf("cold share   plain     nt   uncached   (G lines/s, all gathers; one 1024-thread workgroup per CU)\n");
  for (int c8 : {0, 2, 4, 6, 8}) {
    const float tp = best_of([&] { k_mixed<false><<<grid, 1024>>>(hot, hr, cold, cr, steps, c8, out); });
    const float tn = best_of([&] { k_mixed<true><<<grid, 1024>>>(hot, hr, cold, cr, steps, c8, out); });
    float tu = 0.f;
    if (cold_uc) tu = best_of([&] { k_mixed<false><<<grid, 1024>>>(hot, hr, cold_uc, cr, steps, c8, out); });
    printf("   %d / 8   %7.1f %7.1f %7.1f\n", c8, lines / (tp * 1e-3) / 1e9, lines / (tn * 1e-3) / 1e9,
           cold_uc ? lines / (tu * 1e-3) / 1e9 : 0.0);
  }
  return 0;
}
