// kernels_sort.hip — the index entries of All-Pair-Backward-Search ordered by (source, target) on the device.
//
// Base_Whole_Graph.preprocessing keeps, per source v, a list of (t, pi(v, t)) in target-iteration order
// (Base_Whole_Graph.java:76-92: targets are visited in id order, rows are LinkedHashMaps) and then applies its k rule per
// row (:112-163).  The searches emit their entries as they finish, in no order; bringing 3e7 of them into row order was
// a third of the call on the host (two-level counting sort on 16 threads: 0.4 s at R-MAT 22).  Here the records are
// turned into 64-bit keys source << 32 | target with the value as payload and radix-sorted on the device (rocPRIM; a
// library sort - this is not one of the path's hot kernels) over just the bits the ids use; the host receives rows
// that are contiguous and already in target order and only has to apply the k rule.
#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "device_utils.hpp"
#include "engine.hpp"

namespace pprhip {

__global__ __launch_bounds__(256) void k_rec_to_kv(const TripleRec* __restrict__ rec, unsigned long long count,
                                                    unsigned long long* __restrict__ keys, double* __restrict__ vals) {
  for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * 256ull) {
    const TripleRec r = rec[i];
    keys[i] = ((unsigned long long)(uint32_t)r.v << 32) | (unsigned long long)(uint32_t)r.t;
    vals[i] = r.p;
  }
}

// rec[0 .. count) -> *keys_out / *vals_out (device arrays of `count` entries, owned by the caller: hipFree), ordered
// by (source, target)
int sort_triples_device(pprhip_graph* g, const TripleRec* rec, unsigned long long count, unsigned long long** keys_out,
                        double** vals_out) {
  *keys_out = nullptr;
  *vals_out = nullptr;
  if (count == 0) return PPRHIP_OK;
  if (count >= (1ull << 31)) {
    set_error("index sort: %llu entries exceed the 2^31 the device sort takes", count);
    return PPRHIP_ERR_INVALID;
  }
  unsigned long long *k_in = nullptr, *k_out = nullptr;
  double *v_in = nullptr, *v_out = nullptr;
  void* tmp = nullptr;
  auto fail = [&](int rc) {
    void* p[] = {k_in, k_out, v_in, v_out, tmp};
    for (void* q : p)
      if (q) (void)hipFree(q);
    return rc;
  };
  auto dev = [&](void** p, size_t bytes) -> int {
    const hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
      set_error("index sort: hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? PPRHIP_ERR_OOM : PPRHIP_ERR_HIP;
    }
    return PPRHIP_OK;
  };
  int rc;
  if ((rc = dev((void**)&k_in, 8 * count)) || (rc = dev((void**)&k_out, 8 * count)) || (rc = dev((void**)&v_in, 8 * count)) ||
      (rc = dev((void**)&v_out, 8 * count)))
    return fail(rc);
  const uint32_t grid = (uint32_t)std::min<unsigned long long>((count + 255) / 256, 8192ull);
  k_rec_to_kv<<<dim3(grid), dim3(256), 0, g->stream>>>(rec, count, k_in, v_in);
  if (hipGetLastError() != hipSuccess) return fail(PPRHIP_ERR_HIP);
  // only the bits that ids of this graph can set take part in the sort
  unsigned id_bits = 1;
  while (id_bits < 32 && (1ull << id_bits) < (unsigned long long)g->n) ++id_bits;
  const unsigned end_bit = 32 + id_bits;
  size_t tmp_bytes = 0;
  if (rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)count, 0u, end_bit, g->stream) !=
      hipSuccess) {
    set_error("index sort: sizing the device sort failed");
    return fail(PPRHIP_ERR_HIP);
  }
  if ((rc = dev(&tmp, std::max<size_t>(tmp_bytes, 16)))) return fail(rc);
  if (rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)count, 0u, end_bit, g->stream) != hipSuccess ||
      hipStreamSynchronize(g->stream) != hipSuccess) {
    set_error("index sort: the device sort failed");
    return fail(PPRHIP_ERR_HIP);
  }
  (void)hipFree(k_in);
  (void)hipFree(v_in);
  (void)hipFree(tmp);
  *keys_out = k_out;
  *vals_out = v_out;
  return PPRHIP_OK;
}

int init_kernels_sort() {  // loads this file's code object on the current device (see init_kernels_push)
  hipFuncAttributes fa;
  PPRHIP_CHECK_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&k_rec_to_kv)));
  return PPRHIP_OK;
}

}  // namespace pprhip
