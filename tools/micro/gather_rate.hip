// gather_rate.hip — developer micro-benchmark: rate of random 8-byte gathers on MI355X as a function of the table
// size (L2 / Infinity Cache / HBM), independent loads vs a dependent chain (the walk kernel's shape).
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate tools/micro/gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// every lane: `steps` gathers, `ilp` independent chains in flight
template <int ILP>
__global__ __launch_bounds__(256) void k_gather(const unsigned long long* __restrict__ tab, uint32_t mask, int steps,
                                                 unsigned long long* out) {
  uint32_t s[ILP];
  unsigned long long acc = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s[i] = mix(blockIdx.x * 256u + threadIdx.x + 0x9e3779b9u * (i + 1));
  for (int t = 0; t < steps; ++t) {
    unsigned long long v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = tab[s[i] & mask];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      acc += v[i];
      s[i] = mix(s[i] + (uint32_t)v[i]);  // dependent on the loaded value (a walk's next vertex)
    }
  }
  if (acc == 0x1234567ull) out[0] = acc;
}

int main() {
  const size_t max_elems = (size_t)1 << 28;  // 2 GB of 8-byte entries
  unsigned long long* tab;
  unsigned long long* out;
  hipMalloc(&tab, max_elems * 8);
  hipMalloc(&out, 8);
  std::vector<unsigned long long> h((size_t)1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = i * 2654435761ull;
  for (size_t off = 0; off < max_elems; off += h.size()) hipMemcpy(tab + off, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * 8, steps = 64;
  printf("table_MB ilp gathers_G_per_s\n");
  for (int lg = 19; lg <= 28; ++lg) {
    const uint32_t mask = (uint32_t)(((size_t)1 << lg) - 1);
    for (int ilp : {1, 4}) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (ilp == 1) k_gather<1><<<grid, 256>>>(tab, mask, steps * 4, out);
        else k_gather<4><<<grid, 256>>>(tab, mask, steps, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double n = (double)grid * 256 * steps * 4;
      printf("%8.1f %d %8.1f\n", (double)((size_t)8 << lg) / 1e6, ilp, n / (best * 1e-3) / 1e9);
    }
  }
  return 0;
}
