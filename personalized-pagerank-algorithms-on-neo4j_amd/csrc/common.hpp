// common.hpp — shared declarations of the pprhip engine (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/pprhip.h"

namespace pprhip {

void set_error(const char* fmt, ...);
const char* get_error();

#define PPRHIP_CHECK_HIP(expr)                                                                     \
  do {                                                                                             \
    hipError_t _e = (expr);                                                                        \
    if (_e != hipSuccess) {                                                                        \
      ::pprhip::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return PPRHIP_ERR_HIP;                                                                       \
    }                                                                                              \
  } while (0)

#define PPRHIP_TRY(expr)       \
  do {                         \
    int _rc = (expr);          \
    if (_rc != PPRHIP_OK) return _rc; \
  } while (0)

// Layout constants shared by graph build (host) and kernels (device).
constexpr int kChunkPad = 512;  // in-edges one wave of the dense pull sweep owns (8 per lane)
constexpr int kBlock = 256;

// Batched forward sweep, source-partitioned (engine_internal.hpp: HostPartLayout): the in-edges are cut into kParts
// groups by the partition of their SOURCE id, 64 consecutive ids (8 KB of contribution lines) to a partition in turn.
constexpr int kParts = 8;       // = XCDs of an MI355X: workgroup b of the edge kernel walks partition b % 8
constexpr int kPartShift = 6;
__host__ __device__ inline uint32_t part_of(uint32_t id) { return (id >> kPartShift) & (uint32_t)(kParts - 1); }
// position of an id among the ids of its own partition
__host__ __device__ inline uint32_t part_local(uint32_t id) {
  return ((id >> (kPartShift + 3)) << kPartShift) | (id & ((1u << kPartShift) - 1u));
}
// ... and back: the id at position `local` of partition p
__host__ __device__ inline uint32_t part_global(uint32_t local, uint32_t p) {
  return ((local >> kPartShift) << (kPartShift + 3)) | (p << kPartShift) | (local & ((1u << kPartShift) - 1u));
}
constexpr int kTileRows = 64;   // rows per tile of the batched apply kernel (kernels_push.hip: kApplyRows)

// packed frontier counter: entries in the high 28 bits, edge total in the low 36 bits
constexpr int kPackShift = 36;
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;

}  // namespace pprhip
