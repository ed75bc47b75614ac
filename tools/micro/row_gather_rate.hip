// row_gather_rate.hip — developer micro-benchmark: rate of random ROW gathers on MI355X against the row width.
// W / 8 consecutive lanes read one W-byte row (8 bytes per lane, one coalesced segment) at a random row index of a
// table far beyond L2; 8 independent rows in flight per lane group.  Answers: does a wider row (more queries per
// batched sweep: kBatch * 8 bytes per vertex) raise the bytes per second the cold gathers deliver?
//   hipcc --offload-arch=gfx950 -O3 -o row_gather_rate tools/micro/row_gather_rate.hip && ./row_gather_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int G>  // lanes per row
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ tab, uint32_t row_mask, int steps, double* out) {
  const uint32_t gid = (blockIdx.x * 256u + threadIdx.x) / G, sub = threadIdx.x % G;
  double acc = 0.0;
  uint32_t s = mix(gid + 0x9e3779b9u);
  for (int t = 0; t < steps; ++t) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t r = mix(s + 0x85ebca6bu * (uint32_t)(i + 1)) & row_mask;
      v[i] = tab[(size_t)r * G + sub];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc;
}

template <int G>
static void run(const double* tab, double* out, size_t table_bytes) {
  const uint32_t rows = (uint32_t)(table_bytes / (8 * G));
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grid = 256 * 8, steps = 32;
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    k_rows<G><<<grid, 256>>>(tab, rows - 1, steps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  const double n_rows = (double)grid * 256 / G * steps * 8;
  printf("%8.0f %6d %10.2f %10.1f\n", (double)table_bytes / 1e6, 8 * G, n_rows / (best * 1e-3) / 1e9,
         n_rows * 8 * G / (best * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = (size_t)1 << 31;
  double* tab;
  double* out;
  hipMalloc(&tab, bytes);
  hipMalloc(&out, 8);
  hipMemset(tab, 0, bytes);
  printf("table_MB row_B rows_G_per_s GB_per_s\n");
  for (size_t tb : {(size_t)1 << 29, (size_t)1 << 31}) {
    run<1>(tab, out, tb);
    run<2>(tab, out, tb);
    run<4>(tab, out, tb);
    run<8>(tab, out, tb);
    run<16>(tab, out, tb);
    run<32>(tab, out, tb);
    run<64>(tab, out, tb);
  }
  return 0;
}
