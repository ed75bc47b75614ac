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
// groups by the partition of their SOURCE id, id mod kParts - line by line in turn, so that every partition gets the same
// share of the hot ids (runs of 64 ids gave partition 0, which held the 64 hottest ids, 15 % more edges than the
// average, and the launch waits for its slowest partition; by single lines: 1.3 %).  The L2's channel selection does
// not mind the stride of eight lines (tools/micro/xcd_affine_rate.hip: 102 G lines/s by single lines, 91-94 by runs of
// 8 or 64).
constexpr int kParts = 8;       // = XCDs of an MI355X: workgroup b of the edge kernel walks partition b % 8
__host__ __device__ inline uint32_t part_of(uint32_t id) { return id & (uint32_t)(kParts - 1); }
// Rows of at most kPartWholeRow in-edges are not cut: all their edges go to the partition of their row ordinal (one
// segment, one partial line) - on R-MAT 22 that leaves 4.8 M segments of 7.7 M for 8 % of the edges gathered off
// their source's partition.
constexpr uint32_t kPartWholeRow = 16;
// The copy is a sliced ELL (round 6): a (row, partition) segment is cut into PIECES of at most kPieceMax edges (256
// left the hub rows' block with one slice of 4 096 edges per wave: a serial chain of 64 gather latencies and no way to
// balance the waves), the
// pieces of a group of kGroupRows consecutive rows in one partition are sorted by length and packed sixteen at a time
// into SLICES - one quad of lanes per piece, the slice as wide as its longest piece.
constexpr uint32_t kPieceMax = 64;
constexpr uint32_t kGroupRows = 256;  // = the granularity of the Gauss-Seidel blocks' row boundaries
constexpr int kSliceQuads = 16;       // pieces per slice = quads of a wave
constexpr int kTileRows = 64;   // rows per tile of the batched apply kernel (kernels_push.hip: kApplyRows)

// packed frontier counter: entries in the high 28 bits, edge total in the low 36 bits
constexpr int kPackShift = 36;
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;

}  // namespace pprhip
