// xcd_affine_rate.hip — developer micro-benchmark (round 5) for the two premises of an XCD-affine batched sweep:
//  (1) 128-byte row gathers (16 lanes x 8 bytes: one c8[v][0..15] line) from a table of S MB, either "shared" (every
//      workgroup gathers anywhere: all eight L2s compete for the same S MB) or "affine" (workgroup b gathers only from
//      slice b % 8 of the table: each XCD's L2 sees S / 8 MB, if workgroups are dealt out to the XCDs round-robin);
//      the XCC_ID hardware register is read to check that dealing;
//  (2) what the partial sums of (row, source block) segments would cost: 16 lanes adding to the 16 doubles of a random
//      row of a 256 MB table - as no-return fp64 atomics, as plain row stores, as load + add + store.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o xcd_affine_rate tools/micro/xcd_affine_rate.hip && ./xcd_affine_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void k_xcc(uint32_t* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

// rows_per_slice: rows of one slice; affine: the slice is blockIdx % 8, shared: the row index runs over 8 slices
template <bool AFFINE>
__global__ __launch_bounds__(256) void k_rows(const double* __restrict__ tab, uint32_t rows_per_slice, int steps, double* out) {
  const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) / 16u, sub = threadIdx.x % 16u;
  const uint32_t slice = blockIdx.x % 8u;
  double acc = 0.0;
  uint32_t s = mix(grp + 0x9e3779b9u);
  for (int t = 0; t < steps; ++t) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = mix(s + 0x85ebca6bu * (uint32_t)(i + 1));
      const uint32_t r = AFFINE ? slice * rows_per_slice + h % rows_per_slice : h % (8u * rows_per_slice);
      v[i] = tab[(size_t)r * 16u + sub];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc;
}

// affine by interleaving: XCD x gathers the rows r with ((r >> SH) & 7) == x of the whole table (SH = 0: every 8th line,
// 3: runs of 8 lines = 1 KB, 6: runs of 64 lines = 8 KB) - does the L2's channel selection mind the stride?
template <int SH>
__global__ __launch_bounds__(256) void k_rows_il(const double* __restrict__ tab, uint32_t rows_per_slice, int steps, double* out) {
  const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) / 16u, sub = threadIdx.x % 16u;
  const uint32_t slice = blockIdx.x % 8u;
  double acc = 0.0;
  uint32_t s = mix(grp + 0x9e3779b9u);
  for (int t = 0; t < steps; ++t) {
    double v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const uint32_t h = mix(s + 0x85ebca6bu * (uint32_t)(i + 1)) % rows_per_slice;  // position inside the slice
      const uint32_t r = ((h >> SH) << (SH + 3)) | (slice << SH) | (h & ((1u << SH) - 1u));
      v[i] = tab[(size_t)r * 16u + sub];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc;
}

// the shapes of the two batched edge kernels, one 1024-thread workgroup per CU (LDS bytes of dynamic shared memory keep a
// second one off the CU), XCD-affine rows of a 32 MB table (every 8th line):
//   LPR = 16: sixteen lanes per row, 8 bytes each (k_dense_edges_b), IN rows in flight per lane group
//   LPR = 4:  four lanes per row, 2 x 16 bytes each (k_dense_edges_q)
template <int LPR, int IN>
__global__ __launch_bounds__(1024) void k_shape(const double* __restrict__ tab, uint32_t rows_per_slice, int steps, double* out) {
  extern __shared__ double s_pad[];
  typedef double v2d __attribute__((ext_vector_type(2)));
  const uint32_t grp = (blockIdx.x * 1024u + threadIdx.x) / LPR, sub = threadIdx.x % LPR;
  const uint32_t slice = blockIdx.x % 8u;
  double acc = 0.0;
  uint32_t s = mix(grp + 0x9e3779b9u);
  if (threadIdx.x == 0) s_pad[0] = 0.0;
  for (int t = 0; t < steps; ++t) {
    if (LPR == 16) {
      double v[IN];
#pragma unroll
      for (int i = 0; i < IN; ++i) {
        const uint32_t h = mix(s + 0x85ebca6bu * (uint32_t)(i + 1)) % rows_per_slice;
        v[i] = tab[(size_t)((h << 3) | slice) * 16u + sub];
      }
#pragma unroll
      for (int i = 0; i < IN; ++i) acc += v[i];
    } else {
      v2d a[IN], b[IN];
#pragma unroll
      for (int i = 0; i < IN; ++i) {
        const uint32_t h = mix(s + 0x85ebca6bu * (uint32_t)(i + 1)) % rows_per_slice;
        const v2d* p = reinterpret_cast<const v2d*>(tab + (size_t)((h << 3) | slice) * 16u + 4u * sub);
        a[i] = p[0];
        b[i] = p[1];
      }
#pragma unroll
      for (int i = 0; i < IN; ++i) acc += a[i].x + a[i].y + b[i].x + b[i].y;
    }
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc + s_pad[0];
}

enum { kAtomic = 0, kStore, kLoadAddStore };
template <int KIND>
__global__ __launch_bounds__(256) void k_row_update(double* __restrict__ tab, uint32_t rows, int steps, double* out) {
  const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) / 16u, sub = threadIdx.x % 16u;
  uint32_t s = mix(grp + 0x9e3779b9u);
  double acc = 0.0;
  for (int t = 0; t < steps; ++t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t r = mix(s + 0x85ebca6bu * (uint32_t)(i + 1)) % rows;
      double* p = &tab[(size_t)r * 16u + sub];
      if (KIND == kAtomic) (void)__hip_atomic_fetch_add(p, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (KIND == kStore) *p = (double)t;
      if (KIND == kLoadAddStore) *p = *p + 1.0;
    }
    s = mix(s + 1u);
  }
  if (acc == 0.12345) out[0] = acc;
}

template <class F>
static float best_of(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  double *tab, *out;
  uint32_t* xcc;
  hipMalloc(&tab, bytes);
  hipMalloc(&out, 8);
  hipMalloc(&xcc, 4096 * 4);
  hipMemset(tab, 0, bytes);
  // --- the dealing of workgroups to XCDs
  k_xcc<<<4096, 64>>>(xcc);
  std::vector<uint32_t> h(4096);
  hipMemcpy(h.data(), xcc, 4096 * 4, hipMemcpyDeviceToHost);
  int match = 0;
  for (int i = 0; i < 4096; ++i) match += (h[i] == (uint32_t)(i % 8));
  printf("XCC_ID == blockIdx %% 8 for %d of 4096 workgroups; first 16:", match);
  for (int i = 0; i < 16; ++i) printf(" %u", h[i]);
  printf("\n");
  // --- (1) row gathers
  const int grid = 256 * 8, steps = 64;
  const double n_rows = (double)grid * 256 / 16 * steps * 8;
  printf("table_MB   shared_Glines/s   affine_Glines/s   (128-byte rows; affine: each XCD in table/8)\n");
  for (size_t mb : {2, 4, 8, 16, 24, 32, 48, 64, 128, 256, 1024}) {
    const uint32_t rps = (uint32_t)((mb << 20) / 128 / 8);
    const float t_sh = best_of([&] { k_rows<false><<<grid, 256>>>(tab, rps, steps, out); });
    const float t_af = best_of([&] { k_rows<true><<<grid, 256>>>(tab, rps, steps, out); });
    printf("%8zu %17.1f %17.1f\n", mb, n_rows / (t_sh * 1e-3) / 1e9, n_rows / (t_af * 1e-3) / 1e9);
  }
  printf("table_MB   interleaved affine, runs of 1 / 8 / 64 lines (G lines/s)\n");
  for (size_t mb : {16, 32, 64, 128}) {
    const uint32_t rps = (uint32_t)((mb << 20) / 128 / 8);
    const float t0 = best_of([&] { k_rows_il<0><<<grid, 256>>>(tab, rps, steps, out); });
    const float t3 = best_of([&] { k_rows_il<3><<<grid, 256>>>(tab, rps, steps, out); });
    const float t6 = best_of([&] { k_rows_il<6><<<grid, 256>>>(tab, rps, steps, out); });
    printf("%8zu %10.1f %10.1f %10.1f\n", mb, n_rows / (t0 * 1e-3) / 1e9, n_rows / (t3 * 1e-3) / 1e9, n_rows / (t6 * 1e-3) / 1e9);
  }
  {
    const uint32_t rps = (uint32_t)(((size_t)32 << 20) / 128 / 8);
    const int g1 = 256, st = 64;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shape<16, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shape<4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shape<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shape<16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int lds_kb : {0, 128}) {
      const float tb = best_of([&] { k_shape<16, 8><<<g1, 1024, lds_kb * 1024>>>(tab, rps, st, out); });
      const float tb16 = best_of([&] { k_shape<16, 16><<<g1, 1024, lds_kb * 1024>>>(tab, rps, st, out); });
      const float tq4 = best_of([&] { k_shape<4, 4><<<g1, 1024, lds_kb * 1024>>>(tab, rps, st, out); });
      const float tq8 = best_of([&] { k_shape<4, 8><<<g1, 1024, lds_kb * 1024>>>(tab, rps, st, out); });
      const double rb = (double)g1 * 1024 / 16 * st * 8, rb16 = (double)g1 * 1024 / 16 * st * 16;
      const double rq4 = (double)g1 * 1024 / 4 * st * 4, rq8 = (double)g1 * 1024 / 4 * st * 8;
      printf("one 1024-thread workgroup per CU, %3d KB LDS, affine 32 MB: 16 lanes x 8 B, 8 in flight %.1f; 16 in flight %.1f; "
             "4 lanes x 32 B, 4 in flight %.1f; 8 in flight %.1f G lines/s\n", lds_kb, rb / (tb * 1e-3) / 1e9,
             rb16 / (tb16 * 1e-3) / 1e9, rq4 / (tq4 * 1e-3) / 1e9, rq8 / (tq8 * 1e-3) / 1e9);
    }
    // and with four such workgroups' worth of waves per CU (grid 1024, no LDS): occupancy 16 waves per SIMD if registers allow
    const float tq = best_of([&] { k_shape<4, 4><<<1024, 1024, 0>>>(tab, rps, st, out); });
    printf("grid 1024 x 1024 threads, no LDS: 4 lanes x 32 B, 4 in flight %.1f G lines/s\n", (double)1024 * 1024 / 4 * st * 4 / (tq * 1e-3) / 1e9);
  }
  // --- (2) row updates on a 256 MB table
  const uint32_t rows = (uint32_t)(((size_t)256 << 20) / 128);
  const double n_upd = (double)grid * 256 / 16 * steps * 4;
  const float t_at = best_of([&] { k_row_update<kAtomic><<<grid, 256>>>(tab, rows, steps, out); });
  const float t_st = best_of([&] { k_row_update<kStore><<<grid, 256>>>(tab, rows, steps, out); });
  const float t_ls = best_of([&] { k_row_update<kLoadAddStore><<<grid, 256>>>(tab, rows, steps, out); });
  printf("row updates on 256 MB (G rows/s): 16-lane fp64 atomics %.1f, plain row stores %.1f, load+add+store %.1f\n",
         n_upd / (t_at * 1e-3) / 1e9, n_upd / (t_st * 1e-3) / 1e9, n_upd / (t_ls * 1e-3) / 1e9);
  // and on a small table (4 MB: L2-resident everywhere)
  const uint32_t rows_s = (uint32_t)(((size_t)4 << 20) / 128);
  const float t_at2 = best_of([&] { k_row_update<kAtomic><<<grid, 256>>>(tab, rows_s, steps, out); });
  printf("row updates on 4 MB: 16-lane fp64 atomics %.1f G rows/s\n", n_upd / (t_at2 * 1e-3) / 1e9);
  return 0;
}
