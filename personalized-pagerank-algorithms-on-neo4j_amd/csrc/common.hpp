// common.hpp — shared declarations of the pprhip engine (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

// Environment.  The product library (libpprhip.so) reads the six tuning variables that include/pprhip.h documents and no
// others; fault injection, diagnostics on stderr and the measurement switches of the A/B runs are read by
// libpprhip_hooks.so only (the same sources with -DPPRHIP_TEST_HOOKS: the tests that need them and tools/exp load that).
inline const char* tuning_env(const char* name) { return getenv(name); }
#ifdef PPRHIP_TEST_HOOKS
inline const char* hook_env(const char* name) { return getenv(name); }
#else
inline const char* hook_env(const char*) { return nullptr; }
#endif

// Layout constants shared by graph build (host) and kernels (device).
constexpr int kChunkPad = 512;  // in-edges one wave of the dense pull sweep owns (8 per lane)
constexpr int kBlock = 256;

// Single-query forward sweep over ROW PANELS (engine_internal.hpp: HostPanelLayout, round 6): the rows with in-edges are
// cut into panels of kPanelRows consecutive ordinals whose sums fit a CU's LDS, a panel's in-edges are kept sorted by
// source, and a workgroup sums an ITEM - a part of at most kItemEdges edges of one panel - into accumulators in LDS.
constexpr uint32_t kPanelRows = 8192;    // x 8 B = 64 KB: two workgroups share a CU's 160 KB
constexpr uint32_t kPanelStep = 8192;    // edges a workgroup takes per turn (1024 lanes x 8): items are padded to it
constexpr uint32_t kItemEdges = 32768;   // a panel with more edges is cut into parts of about this many
struct PanelItem {                       // one unit of work of the edge kernel
  uint32_t edge0;                        // first edge, in units of kPanelStep
  uint32_t steps;                        // turns
  uint32_t panel;                        // rows [panel * kPanelRows, ...)
  uint32_t part0;                        // where the item's sums go: part[part0 + local row]
};
constexpr uint32_t kFoldParts = 32;      // a panel of more than kFoldMin parts has them added kFoldParts at a time first
constexpr uint32_t kFoldMin = 16;
constexpr uint32_t kNoFold = 0xffffffffu;
struct PanelDesc {                       // per panel: its S parts' sums are part[base + k * rows + local row], k < S;
  uint32_t base, parts, rows, fold;      // fold != kNoFold: ceil(S / kFoldParts) sums of kFoldParts parts each at
};                                       // part[fold + g * rows + local row] (k_panel_fold)
constexpr int kTileRows = 64;   // rows per tile of the batched apply kernel (kernels_push.hip: kApplyRows)

// packed frontier counter: entries in the high 28 bits, edge total in the low 36 bits
constexpr int kPackShift = 36;
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;

}  // namespace pprhip
