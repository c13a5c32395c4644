// Shared helpers for the gfx950 kernels of libdadet_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/dadet.h"

namespace dadet {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return DADET_ELAUNCH;
  }
  return DADET_OK;
}

#define DADET_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      ::dadet::set_error(__VA_ARGS__);      \
      return DADET_EINVAL;                  \
    }                                       \
  } while (0)

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// MI355X: 256 CUs in 8 XCDs; memory-bound grids are capped and grid-strided (guide G11).
constexpr int kNumCU = 256;
constexpr int kNumXCD = 8;
constexpr int kMaxStreamBlocks = kNumCU * 8;

// Bijective XCD-aware remap of a linear workgroup id (guide T1): hardware places block b on
// XCD b % 8; give each XCD a contiguous chunk of the logical tile space so neighbouring tiles
// (which share operand panels) hit the same L2.
__device__ inline int xcd_remap(int bid, int nwg) {
  const int q = nwg / kNumXCD, r = nwg % kNumXCD;
  const int xcd = bid % kNumXCD, idx = bid / kNumXCD;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace dadet
