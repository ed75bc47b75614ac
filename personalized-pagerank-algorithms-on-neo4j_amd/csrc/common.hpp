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

// Batched forward sweep over ROW PANELS (engine_internal.hpp: HostPartLayout, round 6): the rows with in-edges are cut
// into panels of kPanelRows consecutive ordinals, a panel's in-edges are kept sorted by source, and a workgroup sums a
// panel (or, for the hub rows' panels, a part of at most kItemEdges edges of one) into accumulators in LDS.
constexpr uint32_t kPanelRows = 1024;    // x 16 columns x 8 B = 128 KB of a CU's 160 KB
constexpr uint32_t kPanelStep = 1024;    // edges a workgroup takes per turn (256 quads x 4): items are padded to it
constexpr uint32_t kItemEdges = 32768;   // a panel with more edges is cut into parts of about this many
struct PanelItem {                       // one unit of work of the edge kernel
  uint32_t edge0;                        // first edge, in units of kPanelStep
  uint32_t steps;                        // turns
  uint32_t panel;                        // rows [panel * kPanelRows, ...)
  uint32_t line0;                        // partial line of the panel's first row for this part
  uint32_t stride;                       // parts of its panel = lines between consecutive rows
  uint32_t pad[3];
};
constexpr int kTileRows = 64;   // rows per tile of the batched apply kernel (kernels_push.hip: kApplyRows)

// packed frontier counter: entries in the high 28 bits, edge total in the low 36 bits
constexpr int kPackShift = 36;
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;

}  // namespace pprhip
