// chain_latency.hip - developer micro-benchmark: the latency of ONE dependent random gather (a walk step's record read)
// on an idle chip, against the table size: one wave, every lane its own chain of `steps` dependent 16-byte loads.
//   hipcc --offload-arch=gfx950 -O3 -o chain_latency tools/micro/chain_latency.hip && ./chain_latency
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ void k_fill(uint4* tab, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    tab[i] = make_uint4(mix(i * 2654435761u + 12345u) % n, i, 0, 0);
}
__global__ __launch_bounds__(64) void k_chain(const uint4* __restrict__ tab, uint32_t n, int steps, uint32_t* out) {
  uint32_t idx = mix(blockIdx.x * 64u + threadIdx.x + 17u) % n, acc = 0;
  for (int s = 0; s < steps; ++s) {
    const uint4 r = tab[idx];
    idx = r.x;
    acc += r.y;
  }
  if (acc == 0xdeadbeefu) out[0] = acc;
}
int main() {
  uint32_t* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  printf("table_MB  waves  ns_per_dependent_step\n");
  for (size_t mb : {1, 4, 32, 256, 1024, 4096}) {
    const uint32_t n = (uint32_t)(mb * 1024 * 1024 / 16);
    uint4* tab;
    if (hipMalloc(&tab, (size_t)n * 16) != hipSuccess) break;
    k_fill<<<4096, 256>>>(tab, n);
    for (int waves : {1, 256, 2048}) {
      const int steps = 2000;
      k_chain<<<waves, 64>>>(tab, n, steps, out);
      hipEventRecord(e0);
      k_chain<<<waves, 64>>>(tab, n, steps, out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%8zu  %5d  %8.1f\n", mb, waves, 1e6 * ms / steps);
    }
    hipFree(tab);
  }
  return 0;
}
