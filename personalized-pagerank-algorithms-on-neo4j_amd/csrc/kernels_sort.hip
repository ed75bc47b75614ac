// kernels_sort.hip — the index of All-Pair-Backward-Search finished on the device: rows in the reference's order, the
// k rule applied, before anything crosses PCIe.
//
// Base_Whole_Graph.preprocessing keeps, per source v, a list of (t, pi(v, t)) in target-iteration order
// (Base_Whole_Graph.java:76-92: targets are visited in id order, rows are LinkedHashMaps) and then applies its k rule per
// row (:112-163): k < 0 keeps the row as it is; k >= 0 keeps the entries >= the k-th largest value (all of them when
// the row holds fewer than k), value descending, and - the sort being stable - ties in target order.  The searches emit
// their entries as they finish, in no order.
//   round 1: everything on the host (two-level counting sort on 16 threads: 0.4 s for R-MAT 22's 3e7 entries);
//   round 3: records -> keys source << 32 | target, radix-sorted on the device, the k rule on the host's threads
//            (36 ms at R-MAT 22, 160 ms at R-MAT 24, plus passes over all n rows whatever the entry count: 13 ms for an
//            index of a thousand targets - what a rank of a sharded job pays for its n / 8 targets);
//   now    : the row order comes from three stable radix sorts (rocPRIM; library sorts - not hot kernels of the path),
//            least significant criterion first - by target, by value descending, by source -, so a row is contiguous,
//            value-descending and target-ordered among equal values; the k rule is then a prefix of every row (the row's
//            k-th entry is its k-th largest), kept counts are scanned into the offsets, and the kept entries are
//            scattered into the final targets / values arrays.  The host receives the index arrays themselves
//            (12 bytes per kept entry instead of 16 per found one) and does no per-entry work.
// Every id is checked when the records are turned into keys (source inside the range the caller collected them for,
// target inside [0, n)): the sorts look at the bits ids of this graph can set and nothing else, so a bad id - a
// transport gone wrong, a peer's bad partition - must be an error before them, not an entry in somebody else's row.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

typedef unsigned long long u64;

__global__ __launch_bounds__(256) void k_rec_to_kv(const TripleRec* __restrict__ rec, u64 count, u64* __restrict__ keys,
                                                    u64* __restrict__ vals, uint32_t v_lo, uint32_t v_hi, uint32_t n,
                                                    unsigned int* __restrict__ err) {
  bool bad = false;
  for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (u64)gridDim.x * 256ull) {
    const TripleRec r = rec[i];
    bad |= (uint32_t)r.v < v_lo || (uint32_t)r.v >= v_hi || (uint32_t)r.t >= n;
    keys[i] = ((u64)(uint32_t)r.v << 32) | (u64)(uint32_t)r.t;
    vals[i] = (u64)__double_as_longlong(r.p);
  }
  if (bad) atomicOr(err, 1u);
}

// start[v] = first entry whose source is >= v, for v in [0, n] (the entries are ordered by source)
__global__ __launch_bounds__(256) void k_row_starts(const u64* __restrict__ keys, u64 count, uint32_t n,
                                                     u64* __restrict__ start) {
  const u64 v = blockIdx.x * 256ull + threadIdx.x;
  if (v > n) return;
  u64 lo = 0, hi = count;
  while (lo < hi) {
    const u64 mid = (lo + hi) >> 1;
    if ((keys[mid] >> 32) < v) lo = mid + 1;
    else hi = mid;
  }
  start[v] = lo;
}

// kept[v] = entries row v keeps (Base_Whole_Graph.java:112-163); rows are value-descending when k >= 0, so the entries
// >= the k-th largest are the first k and whatever equals the k-th behind them.  kept[n] = 0 (the scan's last input).
__global__ __launch_bounds__(256) void k_row_kept(const u64* __restrict__ start, const u64* __restrict__ vals,
                                                   uint32_t n, int k, u64* __restrict__ kept) {
  const u64 v = blockIdx.x * 256ull + threadIdx.x;
  if (v > n) return;
  if (v == n) {
    kept[v] = 0;
    return;
  }
  const u64 b = start[v], len = start[v + 1] - b;
  u64 c = len;
  if (k >= 1 && (u64)k <= len) {  // (fewer than k entries, or k = 0: kth_ppr returns null and everything stays, :133-139)
    const double kth = __longlong_as_double((long long)vals[b + k - 1]);
    c = (u64)k;
    while (c < len && __longlong_as_double((long long)vals[b + c]) >= kth) ++c;
  }
  kept[v] = c;
}

__global__ __launch_bounds__(256) void k_emit_rows(const u64* __restrict__ keys, const u64* __restrict__ vals, u64 count,
                                                    const u64* __restrict__ start, const u64* __restrict__ offsets,
                                                    int32_t* __restrict__ targets, double* __restrict__ values) {
  for (u64 i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (u64)gridDim.x * 256ull) {
    const u64 key = keys[i];
    const uint32_t v = (uint32_t)(key >> 32);
    const u64 j = i - start[v], o = offsets[v];
    if (j < offsets[v + 1] - o) {
      targets[o + j] = (int32_t)(uint32_t)key;
      values[o + j] = __longlong_as_double((long long)vals[i]);
    }
  }
}

void device_rows_free(DeviceRows* r) {
  if (r->offsets) (void)hipFree(r->offsets);
  if (r->targets) (void)hipFree(r->targets);
  if (r->values) (void)hipFree(r->values);
  *r = DeviceRows();
}

// rec[0 .. count) -> the index of the rows of sources [v_lo, v_hi) in device memory: offsets[n + 1], and the kept
// entries' targets / values row by row.  count == 0 leaves all three null (the caller's index is all-empty rows).
int finalize_rows_device(pprhip_graph* g, const TripleRec* rec, u64 count, int k, uint32_t v_lo, uint32_t v_hi,
                         DeviceRows* out) {
  *out = DeviceRows();
  if (count == 0) return PPRHIP_OK;
  const uint32_t n = g->n;
  if (count >= (1ull << 31)) {
    set_error("index: %llu entries exceed the 2^31 the device sort takes", count);
    return PPRHIP_ERR_INVALID;
  }
  u64 *kA = nullptr, *kB = nullptr, *vA = nullptr, *vB = nullptr, *start = nullptr, *kept = nullptr;
  unsigned int* err = nullptr;
  void* tmp = nullptr;
  DeviceRows R;
  auto fail = [&](int rc) {
    void* p[] = {kA, kB, vA, vB, start, kept, err, tmp};
    for (void* q : p)
      if (q) (void)hipFree(q);
    device_rows_free(&R);
    return rc;
  };
  auto dev = [&](void** p, size_t bytes) -> int {
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      set_error("index: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? PPRHIP_ERR_OOM : PPRHIP_ERR_HIP;
    }
    return PPRHIP_OK;
  };
  const size_t rows_bytes = 8 * ((size_t)n + 1);
  int rc;
  if ((rc = dev((void**)&kA, 8 * count)) || (rc = dev((void**)&kB, 8 * count)) || (rc = dev((void**)&vA, 8 * count)) ||
      (rc = dev((void**)&vB, 8 * count)) || (rc = dev((void**)&start, rows_bytes)) || (rc = dev((void**)&kept, rows_bytes)) ||
      (rc = dev((void**)&R.offsets, rows_bytes)) || (rc = dev((void**)&err, sizeof(unsigned int))))
    return fail(rc);
  if (hipMemsetAsync(err, 0, sizeof(unsigned int), g->stream) != hipSuccess) return fail(PPRHIP_ERR_HIP);
  const uint32_t grid = (uint32_t)std::min<u64>((count + 255) / 256, 8192ull);
  k_rec_to_kv<<<dim3(grid), dim3(256), 0, g->stream>>>(rec, count, kA, vA, v_lo, v_hi, n, err);
  if (hipGetLastError() != hipSuccess) return fail(PPRHIP_ERR_HIP);
  // only the bits that ids of this graph can set take part in the sorts by id
  unsigned id_bits = 1;
  while (id_bits < 32 && (1ull << id_bits) < (u64)n) ++id_bits;
  rocprim::double_buffer<u64> dk(kA, kB), dv(vA, vB);
  // k < 0: one sort by (source, target).  k >= 0: by target, then by value descending (the key is the value's bit
  // pattern - the values are positive doubles -, the record's ids ride along), then by source; every sort is stable.
  struct Pass {
    int what;  // 0: ids by bits [lo, hi) ascending; 1: values descending
    unsigned lo, hi;
  };
  const Pass by_row[1] = {{0, 0u, 32u + id_bits}};
  const Pass by_rule[3] = {{0, 0u, id_bits}, {1, 0u, 64u}, {0, 32u, 32u + id_bits}};
  const Pass* passes = k < 0 ? by_row : by_rule;
  const int n_pass = k < 0 ? 1 : 3;
  auto sort_pass = [&](const Pass& p, void* t, size_t& bytes) {
    return p.what == 0 ? rocprim::radix_sort_pairs(t, bytes, dk, dv, (size_t)count, p.lo, p.hi, g->stream)
                       : rocprim::radix_sort_pairs_desc(t, bytes, dv, dk, (size_t)count, p.lo, p.hi, g->stream);
  };
  size_t tmp_bytes = 16;
  for (int i = 0; i < n_pass; ++i) {
    size_t b = 0;
    if (sort_pass(passes[i], nullptr, b) != hipSuccess) {
      set_error("index: sizing the device sort failed");
      return fail(PPRHIP_ERR_HIP);
    }
    tmp_bytes = std::max(tmp_bytes, b);
  }
  {
    size_t b = 0;
    if (rocprim::exclusive_scan(nullptr, b, kept, R.offsets, 0ull, (size_t)n + 1, rocprim::plus<u64>(), g->stream) !=
        hipSuccess) {
      set_error("index: sizing the device scan failed");
      return fail(PPRHIP_ERR_HIP);
    }
    tmp_bytes = std::max(tmp_bytes, b);
  }
  if ((rc = dev(&tmp, tmp_bytes))) return fail(rc);
  for (int i = 0; i < n_pass; ++i) {
    size_t b = tmp_bytes;
    if (sort_pass(passes[i], tmp, b) != hipSuccess) {
      set_error("index: the device sort failed");
      return fail(PPRHIP_ERR_HIP);
    }
  }
  const u64* keys = dk.current();
  const u64* vals = dv.current();
  const uint32_t row_grid = (uint32_t)(((u64)n + 1 + 255) / 256);
  k_row_starts<<<dim3(row_grid), dim3(256), 0, g->stream>>>(keys, count, n, start);
  k_row_kept<<<dim3(row_grid), dim3(256), 0, g->stream>>>(start, vals, n, k, kept);
  if (hipGetLastError() != hipSuccess) return fail(PPRHIP_ERR_HIP);
  {
    size_t b = tmp_bytes;
    if (rocprim::exclusive_scan(tmp, b, kept, R.offsets, 0ull, (size_t)n + 1, rocprim::plus<u64>(), g->stream) != hipSuccess) {
      set_error("index: the device scan failed");
      return fail(PPRHIP_ERR_HIP);
    }
  }
  unsigned int h_err = 0;
  u64 total = 0;
  if (hipMemcpyAsync(&h_err, err, sizeof h_err, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
      hipMemcpyAsync(&total, R.offsets + n, sizeof total, hipMemcpyDeviceToHost, g->stream) != hipSuccess ||
      hipStreamSynchronize(g->stream) != hipSuccess) {
    set_error("index: the device finalisation failed");
    return fail(PPRHIP_ERR_HIP);
  }
  if (h_err) {
    set_error("index entry with a source outside [%u, %u) or a target outside [0, %u)", v_lo, v_hi, n);
    return fail(PPRHIP_ERR_INVALID);
  }
  if (total > count) {
    set_error("index: %llu entries kept of %llu", total, count);
    return fail(PPRHIP_ERR_STATE);
  }
  if ((rc = dev((void**)&R.targets, 4 * std::max<u64>(total, 1))) || (rc = dev((void**)&R.values, 8 * std::max<u64>(total, 1))))
    return fail(rc);
  k_emit_rows<<<dim3(grid), dim3(256), 0, g->stream>>>(keys, vals, count, start, R.offsets, R.targets, R.values);
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(g->stream) != hipSuccess) {
    set_error("index: writing the rows failed");
    return fail(PPRHIP_ERR_HIP);
  }
  R.entries = total;
  void* p[] = {kA, kB, vA, vB, start, kept, err, tmp};
  for (void* q : p) (void)hipFree(q);
  *out = R;
  return PPRHIP_OK;
}

int init_kernels_sort() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rec_to_kv)));
  return PPRHIP_OK;
}

}  // namespace pprhip
