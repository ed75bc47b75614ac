// chain_rate.hip — developer micro-benchmark: dependent random 16-byte gathers (a walk step of kernels_walk.hip: the
// record read gives the index of the next read), against chains per lane and waves per CU.
//   table: N uint4 records (1 GB), record i holds a pseudo-random next index
//   every lane follows ILP independent chains for `steps` steps; rate = lanes x ILP x steps / time
//   hipcc --offload-arch=gfx950 -O3 -o chain_rate tools/micro/chain_rate.hip && ./chain_rate
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

template <int ILP>
__global__ __launch_bounds__(64) void k_chain(const uint4* __restrict__ tab, uint32_t n, int steps, uint32_t* out,
                                               int lds_pad) {
  extern __shared__ char pad[];  // occupancy control
  if (lds_pad < 0) pad[0] = 0;
  uint32_t idx[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) idx[i] = mix((blockIdx.x * 64u + threadIdx.x) * ILP + i) % n;
  uint32_t acc = 0;
  for (int s = 0; s < steps; ++s) {
    uint4 r[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) r[i] = tab[idx[i]];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      idx[i] = r[i].x;
      acc += r[i].y;
    }
  }
  if (acc == 0xdeadbeefu) out[0] = acc;
}

int main() {
  const uint32_t n = 1u << 26;  // 64 M records = 1 GB
  uint4* tab;
  uint32_t* out;
  if (hipMalloc(&tab, (size_t)n * 16) != hipSuccess) return 1;
  hipMalloc(&out, 4);
  k_fill<<<4096, 256>>>(tab, n);
  hipDeviceSynchronize();
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_chain<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("%-4s %-12s %-10s %10s %12s\n", "ILP", "waves/CU", "lanes", "G steps/s", "us per step");
  const int steps = 200;
  for (int ilp = 1; ilp <= 4; ilp *= 2)
    for (int wpc : {4, 8, 16, 32}) {
      const int lds = 160 * 1024 / wpc - 512;  // so that exactly wpc one-wave workgroups fit a CU's LDS
      const int lds_use = wpc == 32 ? 0 : (lds > 65536 ? 65536 : lds);
      const int eff = wpc == 32 ? 32 : (lds > 65536 ? (160 * 1024) / 65536 : wpc);
      const uint32_t grid = 256u * eff;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        if (ilp == 1) k_chain<1><<<grid, 64, lds_use>>>(tab, n, steps, out, 0);
        if (ilp == 2) k_chain<2><<<grid, 64, lds_use>>>(tab, n, steps, out, 0);
        if (ilp == 4) k_chain<4><<<grid, 64, lds_use>>>(tab, n, steps, out, 0);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double total = (double)grid * 64 * ilp * steps;
      printf("%-4d %-12d %-10u %10.2f %12.3f\n", ilp, eff, grid * 64, total / (best * 1e-3) / 1e9, best * 1e3 / steps);
      fflush(stdout);
    }
  return 0;
}
