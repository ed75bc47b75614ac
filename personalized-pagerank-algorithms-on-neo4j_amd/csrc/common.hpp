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

constexpr int kTileRows = 64;   // rows per tile of the batched apply kernel (kernels_push.hip: kApplyRows)

// packed frontier counter: entries in the high 28 bits, edge total in the low 36 bits
constexpr int kPackShift = 36;
constexpr unsigned long long kPackMask = (1ull << kPackShift) - 1ull;

}  // namespace pprhip
