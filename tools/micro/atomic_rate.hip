// atomic_rate.hip — developer micro-benchmark: what a random read-modify-write of one double costs on MI355X, in the
// shapes All-Pair's per-target tables can take (kernels_apbs.hip).  Every workgroup updates random entries of a table
// of its own (`per_wg` doubles; the sum over the workgroups in flight is the footprint) or of one shared table:
//   atom_ret     returning fp64 atomic add, agent scope        (global_atomic_add_f64 ... glc: the push's residue update)
//   atom_noret   the same without the return value
//   atom_wg      returning, workgroup scope                    (does the scope change where the atomic executes?)
//   cas_u32      32-bit compare-and-swap that always fails      (the probe-and-claim of an open-addressing insert)
//   plain_rmw    plain load, add, plain store (racy; what a private, conflict-free update would cost)
//   probe+atom   a bypassing 4-byte key load and then the atomic on the same 16-byte slot (a hash-table edge)
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o atomic_rate tools/micro/atomic_rate.hip && ./atomic_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

enum { kAtomRet = 0, kAtomNoRet, kAtomWg, kCas, kPlain, kProbeAtom };

template <int KIND, int ILP>
__global__ __launch_bounds__(1024) void k_rmw(double* __restrict__ tab, size_t wg_stride, uint32_t mask, int steps,
                                               double* out) {
  double* T = tab + (size_t)blockIdx.x * wg_stride;
  uint32_t s[ILP];
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s[i] = mix(blockIdx.x * 1024u + threadIdx.x + 0x9e3779b9u * (i + 1));
  for (int t = 0; t < steps; ++t) {
    double v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      const uint32_t j = s[i] & mask;
      if (KIND == kAtomRet) v[i] = __hip_atomic_fetch_add(&T[j], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (KIND == kAtomNoRet) { (void)__hip_atomic_fetch_add(&T[j], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); v[i] = 0.0; }
      if (KIND == kAtomWg) v[i] = __hip_atomic_fetch_add(&T[j], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (KIND == kCas) v[i] = (double)atomicCAS((int*)&T[j], -7, 1);
      if (KIND == kPlain) { v[i] = T[j]; }
      if (KIND == kProbeAtom) {
        const uint32_t j2 = j & ~1u;  // 16-byte slot {key, residue}
        const int k = __hip_atomic_load((int*)&T[j2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v[i] = (k == -7) ? 0.0 : __hip_atomic_fetch_add(&T[j2 + 1], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
#pragma unroll
    for (int i = 0; i < ILP; ++i) {
      if (KIND == kPlain) T[s[i] & mask] = v[i] + 1.0;
      acc += v[i];
      s[i] = mix(s[i] + 0x9e3779b9u);
    }
  }
  if (acc == -1.0) out[0] = acc;
}

int main(int argc, char** argv) {
  const size_t max_bytes = (size_t)8 << 30;
  double *tab, *out;
  if (hipMalloc(&tab, max_bytes) != hipSuccess) return 1;
  hipMalloc(&out, 8);
  hipMemset(tab, 0, max_bytes);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const char* names[] = {"atom_ret", "atom_noret", "atom_wg", "cas_u32", "plain_rmw", "probe+atom"};
  printf("%-11s %6s %10s %12s %10s\n", "kind", "wgs", "per_wg_KB", "footprint_MB", "G_ops/s");
  struct Cfg { int wgs; size_t per_wg; bool shared; };
  const Cfg cfgs[] = {
      {256, (size_t)1 << 16, false},   // 512 KB per workgroup: 128 MB in flight
      {256, (size_t)1 << 17, false},   // 1 MB: 256 MB
      {256, (size_t)1 << 19, false},   // 4 MB: 1 GB
      {256, (size_t)1 << 22, false},   // 32 MB (a dense residue vector at R-MAT 22): 8 GB
      {64, (size_t)1 << 17, false},    // 64 workgroups x 1 MB = 64 MB
      {24, (size_t)1 << 17, false},    // 3 per XCD x 1 MB (a table set that fits the L2s)
      {256, (size_t)1 << 22, true},    // one shared 32 MB table
      {256, (size_t)1 << 28, true},    // one shared 2 GB table
  };
  for (int kind = 0; kind < 6; ++kind)
    for (const Cfg& c : cfgs) {
      const int steps = 128;
      const uint32_t mask = (uint32_t)(c.per_wg - 1);
      const size_t stride = c.shared ? 0 : c.per_wg;
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        switch (kind) {
          case kAtomRet: k_rmw<kAtomRet, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
          case kAtomNoRet: k_rmw<kAtomNoRet, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
          case kAtomWg: k_rmw<kAtomWg, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
          case kCas: k_rmw<kCas, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
          case kPlain: k_rmw<kPlain, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
          default: k_rmw<kProbeAtom, 4><<<c.wgs, 1024>>>(tab, stride, mask, steps, out); break;
        }
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      const double n = (double)c.wgs * 1024 * steps * 4;
      printf("%-11s %6d %10.0f %12.0f %10.2f%s\n", names[kind], c.wgs, c.per_wg * 8.0 / 1024,
             (c.shared ? 1.0 : (double)c.wgs) * c.per_wg * 8.0 / 1e6, n / (best * 1e-3) / 1e9, c.shared ? "  (shared)" : "");
      fflush(stdout);
    }
  return 0;
}
