// kernels_select.hip — k-th largest / top-k over the reserve vector (Algo_Util.kth_ppr,
// Algo_Util.java:32-53, and retrieveTopK, Fora_Topk.java:186-199).
//
// The reference quickselects over the values of its sparse map; here the estimates sit in a dense
// fp64 vector in HBM, entries > 0 are the map's entries, and positive doubles order like their bit
// patterns, so a radix select over the 64-bit patterns (12-bit digits, LDS histograms) finds the
// k-th value in a few 8n-byte streaming passes; a final pass gathers every entry >= that value.
#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

constexpr int kHistBins = 4096;

// AGG (the first pass, whose digit is sign + exponent): the entries of a query's estimate lie within a few dozen
// exponents, so the 64 lanes of a wave hit a handful of bins and their LDS atomics on one address are served one after
// the other; here the lanes that hold the leader's bin are counted with a ballot and the leader adds them at once - a
// few trips per wave instead of up to 64 (round 5: 21 -> ~9 us per pass over R-MAT 22's live range).  Later passes
// (mantissa digits: uniform bins) keep one atomic per lane.  A thread takes 4 consecutive entries per trip (two
// 16-byte loads in flight).
template <bool AGG>
__global__ __launch_bounds__(256) void k_select_hist(const double* __restrict__ x, uint32_t n,
                                                      unsigned long long prefix, int prefix_bits, int digit_bits,
                                                      uint32_t* __restrict__ hist, unsigned long long* zero_word) {
  __shared__ uint32_t s_hist[kHistBins];
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0ull;  // the candidate counter of the gather to come
  const int bins = 1 << digit_bits;
  for (int b = threadIdx.x; b < bins; b += blockDim.x) s_hist[b] = 0;
  __syncthreads();
  const int shift = 64 - prefix_bits - digit_bits;
  const int lane = lane_id();
  const uint32_t n4 = (n + 3u) / 4u;  // groups of 4 entries (the array is allocated for n rounded up: engine.cpp)
  for (uint32_t g4 = blockIdx.x * blockDim.x + threadIdx.x; g4 - threadIdx.x % 64u < n4; g4 += gridDim.x * blockDim.x) {
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    if (g4 < n4) {
      if (4u * g4 + 4u <= n) {
        const double2* p = reinterpret_cast<const double2*>(x + 4u * (size_t)g4);
        const double2 a = p[0], b = p[1];
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
      } else {
        for (int i = 0; i < 4; ++i)
          if (4u * g4 + i < n) v[i] = x[4u * (size_t)g4 + i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(v[i]);
      const bool valid = v[i] > 0.0 && !(prefix_bits > 0 && (bits >> (64 - prefix_bits)) != prefix);
      const uint32_t bin = (uint32_t)((bits >> shift) & (unsigned long long)(bins - 1));
      if (AGG) {
        unsigned long long todo = __ballot(valid);
        while (todo) {
          const int leader = __ffsll((long long)todo) - 1;
          const uint32_t lb = (uint32_t)__shfl((int)bin, leader);
          const unsigned long long same = __ballot(valid && bin == lb);
          if (lane == leader) atomicAdd(&s_hist[lb], (uint32_t)__popcll(same));
          todo &= ~same;
        }
      } else if (valid) {
        atomicAdd(&s_hist[bin], 1u);
      }
    }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += blockDim.x) {
    const uint32_t c = s_hist[b];
    if (c) atomicAdd(&hist[b], c);
  }
}

// After the first histogram pass (the 12 leading bits: sign and exponent): the bin that holds the k-th largest entry,
// chosen on the device so that the gather can follow without a host round trip.  Header words: [1] the smallest bit
// pattern the gather takes (the chosen bin's lower edge; 1 = every positive entry when fewer than k exist), [2] how
// many entries that will be, [3] positive entries in all, [4] 1 when there are at least k.
__global__ __launch_bounds__(1024) void k_select_choose(const uint32_t* __restrict__ hist, unsigned long long k,
                                                         unsigned long long* __restrict__ hdr) {
  __shared__ unsigned long long s_cnt[kHistBins];
  __shared__ unsigned long long s_part[1024];
  __shared__ unsigned long long s_total;
  // suffix sums over the bins: bin b's count of entries in bins >= b
  unsigned long long mine = 0;
  for (int j = 0; j < kHistBins / 1024; ++j) {
    const int b = threadIdx.x * (kHistBins / 1024) + j;
    s_cnt[b] = hist[b];
    mine += hist[b];
  }
  // exclusive suffix sums of the 1024 per-thread counts (s_part[t] = entries in the bins of the threads above t):
  // ten doubling steps
  s_part[threadIdx.x] = mine;
  __syncthreads();
  unsigned long long incl = mine;
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned long long add = threadIdx.x + d < 1024 ? s_part[threadIdx.x + d] : 0ull;
    __syncthreads();
    incl += add;
    s_part[threadIdx.x] = incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    hdr[3] = incl;
    s_total = incl;
  }
  __syncthreads();
  s_part[threadIdx.x] = incl - mine;
  __syncthreads();
  const unsigned long long total = s_total;
  if (threadIdx.x == 0 && (total == 0 || k > total)) {
    hdr[1] = 1ull;
    hdr[2] = total;
    hdr[4] = 0ull;
  }
  if (total == 0 || k > total) return;
  unsigned long long above = s_part[threadIdx.x];
  for (int j = kHistBins / 1024 - 1; j >= 0; --j) {
    const int b = threadIdx.x * (kHistBins / 1024) + j;
    const unsigned long long c = s_cnt[b];
    if (above < k && above + c >= k) {  // exactly one bin: the k-th largest lies in it
      hdr[1] = (unsigned long long)b << 52;
      hdr[2] = above + c;
      hdr[4] = 1ull;
    }
    above += c;
  }
}

// Candidates are 16-byte records behind a 64-byte header {count, lower bits, expected, total, have k, the round's
// residue sum (sum_cell given: top-k rounds)}: one copy
// brings the header and the records to the host.  lower_dev: take the lower bound from the header (k_select_choose
// wrote it).  Block 0 leaves the histogram of the select that has just been read out all-zero for the next one.
__global__ __launch_bounds__(256) void k_select_gather(const double* __restrict__ x, uint32_t n,
                                                        unsigned long long lower_bits, int lower_dev,
                                                        char* __restrict__ blob, uint32_t cap,
                                                        uint32_t* __restrict__ hist, const double* sum_cell) {
  unsigned long long* count = reinterpret_cast<unsigned long long*>(blob);
  SelRec* recs = reinterpret_cast<SelRec*>(blob + kSelHeader);
  if (lower_dev) lower_bits = count[1];
  if (blockIdx.x == 0) {
    for (int b = threadIdx.x; b < kHistBins; b += blockDim.x) hist[b] = 0u;
    if (sum_cell && threadIdx.x == 0) count[5] = (unsigned long long)__double_as_longlong(*sum_cell);
  }
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t nround = (n + stride - 1) / stride * stride;
  const int lane = lane_id();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nround; i += stride) {
    double v = 0.0;
    bool take = false;
    if (i < n) {
      v = x[i];
      take = v > 0.0 && (unsigned long long)__double_as_longlong(v) >= lower_bits;
    }
    const unsigned long long mask = __ballot(take);
    if (mask == 0) continue;
    const int leader = __ffsll((long long)mask) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomic_add_u64(count, (unsigned long long)__popcll(mask));
    base = __shfl(base, leader);
    if (take) {
      const unsigned long long pos = base + (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
      if (pos < cap) recs[pos] = SelRec{(int32_t)i, 0, v};
    }
  }
}

int init_kernels_select() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_select_hist<true>)));
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_select_hist<false>)));
  return PPRHIP_OK;
}

// first_pass: the histogram is all-zero already (the gather of the select before left it so; alloc_workspace zeroes
// it once) and the pass clears the candidate counter for the gather that follows
int launch_select_hist(pprhip_graph* g, const double* x, uint32_t n, unsigned long long prefix, int prefix_bits,
                       int digit_bits, bool first_pass) {
  uint64_t b = ((uint64_t)n + 256 * 8 - 1) / (256 * 8);
  const uint32_t grid = (uint32_t)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
  if (!first_pass) PPRHIP_CHECK_HIP(hipMemsetAsync(g->hist, 0, sizeof(uint32_t) * kHistBins, g->stream));
  if (prefix_bits == 0)
    hipLaunchKernelGGL(k_select_hist<true>, dim3(grid), dim3(256), 0, g->stream, x, n, prefix, prefix_bits, digit_bits,
                       g->hist, first_pass ? reinterpret_cast<unsigned long long*>(g->sel_blob) : nullptr);
  else
    hipLaunchKernelGGL(k_select_hist<false>, dim3(grid), dim3(256), 0, g->stream, x, n, prefix, prefix_bits, digit_bits,
                       g->hist, first_pass ? reinterpret_cast<unsigned long long*>(g->sel_blob) : nullptr);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_select_choose(pprhip_graph* g, unsigned long long k) {
  hipLaunchKernelGGL(k_select_choose, dim3(1), dim3(1024), 0, g->stream, g->hist, k,
                     reinterpret_cast<unsigned long long*>(g->sel_blob));
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

int launch_select_gather(pprhip_graph* g, const double* x, uint32_t n, unsigned long long lower_bits, bool zero_count,
                         bool lower_from_device, const double* sum_cell) {
  uint64_t b = ((uint64_t)n + 255) / 256;
  const uint32_t grid = (uint32_t)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
  if (zero_count) PPRHIP_CHECK_HIP(hipMemsetAsync(g->sel_blob, 0, sizeof(unsigned long long), g->stream));
  hipLaunchKernelGGL(k_select_gather, dim3(grid), dim3(256), 0, g->stream, x, n, lower_bits, lower_from_device ? 1 : 0,
                     g->sel_blob, g->sel_cap, g->hist, sum_cell);
  PPRHIP_CHECK_HIP(hipGetLastError());
  return PPRHIP_OK;
}

}  // namespace pprhip
