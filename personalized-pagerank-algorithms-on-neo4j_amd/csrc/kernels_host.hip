// kernels_host.hip — what the host asks of the device between two kernels of a query (gfx950): clearing several
// ranges in one launch, and handing a few words back without a copy command.
//
// A query is a chain of short kernels with the host deciding in between (which level form comes next, how many walks,
// which histogram bin holds the k-th value).  Each decision used to cost hipMemcpyAsync + hipStreamSynchronize: a blit
// kernel, the completion interrupt and the wake-up of the waiting thread - 14.4 us per round trip, against 2.9 us
// between two dependent kernels (tools/micro/roundtrip.hip, profiles/r03_roundtrip.txt).  k_publish writes the words
// into mapped pinned memory itself and a sequence word after them; the host spins on the sequence word: 6.6 us.
#include <algorithm>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

__global__ __launch_bounds__(256) void k_publish(const unsigned long long* __restrict__ src, uint32_t n_words,
                                                  HostMail* mail, unsigned long long seq) {
  for (uint32_t i = threadIdx.x; i < n_words; i += 256) mail->words[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(&mail->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_clear(ClearList L) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
  for (int r = 0; r < L.n; ++r) {
    char* base = static_cast<char*>(L.p[r]);
    const size_t bytes = L.bytes[r];
    // 16-byte stores over the aligned middle (the ranges are device allocations or 8-byte offsets into them), 8- and
    // 1-byte stores for what is left at the ends
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(base) & 15)) & 15;
    const size_t h = head < bytes ? head : bytes;
    if (t < h) base[t] = 0;
    const size_t quads = (bytes - h) / 16;
    ulonglong2* q = reinterpret_cast<ulonglong2*>(base + h);
    const ulonglong2 z = make_ulonglong2(0ull, 0ull);
    for (size_t i = t; i < quads; i += stride) q[i] = z;
    const size_t done = h + quads * 16;
    if (t < bytes - done) base[done + t] = 0;
  }
}

// A vector moved inside HBM (a finished query's vector into the result store) and a frontier of one node set up, as
// kernels: the runtime's copy and fill commands cost the host 15-30 us each before the next launch is queued
// (kernel-trace gaps behind k_publish, bench.py: stream_occupancy) - on a path whose kernels take that long themselves.
__global__ __launch_bounds__(256) void k_copy_f64(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = __builtin_nontemporal_load(&src[i]);
}

__global__ void k_seed_one(int32_t* __restrict__ F, uint32_t* __restrict__ eoff, int32_t node) {
  F[0] = node;
  eoff[0] = 0u;
}

int launch_copy_f64(pprhip_graph* g, const double* src, double* dst, size_t n) {
  if (!n) return PPRHIP_OK;
  const uint32_t grid = (uint32_t)std::min<size_t>((n + 256 * 8 - 1) / (256 * 8), (size_t)g->n_cus * 8);
  hipLaunchKernelGGL(k_copy_f64, dim3(grid), dim3(256), 0, g->stream, src, dst, n);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_seed_one(pprhip_graph* g, int fbuf, int32_t node) {
  hipLaunchKernelGGL(k_seed_one, dim3(1), dim3(1), 0, g->stream, g->F[fbuf], g->eoff[fbuf], node);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// Does a stream run beside another?  The runtime spreads streams over a few in-order hardware queues, and which
// streams share one depends on what else the process has created.  k_hold keeps the first stream busy for a while,
// k_publish on the second writes a word to the host; if the word arrives while k_hold still runs, the two overlap.
__global__ void k_hold(unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

int launch_hold(hipStream_t stream, unsigned long long ticks) {
  hipLaunchKernelGGL(k_hold, dim3(1), dim3(64), 0, stream, ticks);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int init_kernels_host() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_publish)));
  return PPRHIP_OK;
}

int launch_publish(pprhip_graph* g, const void* src, uint32_t n_words, unsigned long long seq) {
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, g->stream, static_cast<const unsigned long long*>(src), n_words,
                     g->mail_dev, seq);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

// every range starts on an 8-byte boundary
int launch_clear(pprhip_graph* g, const ClearList& L) {
  size_t total = 0;
  for (int r = 0; r < L.n; ++r) total += L.bytes[r];
  if (total == 0) return PPRHIP_OK;
  const size_t blocks = (total / 8 + 256 * 16 - 1) / (256 * 16);
  const uint32_t grid = (uint32_t)std::min<size_t>(std::max<size_t>(blocks, 1), (size_t)g->n_cus * 8);
  hipLaunchKernelGGL(k_clear, dim3(grid), dim3(256), 0, g->stream, L);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
