// Shared device/host helpers for libgsplat_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/gsplat_mi355x.h"

#define GS_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

#define GS_HIP(x)                                 \
  do {                                            \
    hipError_t e__ = (x);                         \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

namespace gs {

constexpr int REC = GS_REC;  // floats per splat record
constexpr int WAVE = 64;

constexpr float ALPHA_MIN = 1.0f / 255.0f;
constexpr float ALPHA_MAX = 0.999f;
constexpr float T_MIN = 1e-4f;

// tile lists longer than this are 'long': sorted by the big-LDS class of gs_isect_tile_sort, and placed first (as a
// block) by gs_raster_tile_order
constexpr int TILE_LONG = 2048;

__host__ __device__ inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// XCD-aware, bijective remap of a linear workgroup id: workgroups b and b+8 share an XCD (observed
// round-robin dispatch), so give each XCD label a contiguous range of work items -> neighbouring
// tiles (which gather the same splat records) hit the same 4 MiB L2.  Speed only, never correctness.
__device__ inline unsigned xcd_remap(unsigned bid, unsigned n) {
  unsigned q = n >> 3, r = n & 7u;
  unsigned xcd = bid & 7u, idx = bid >> 3;
  unsigned base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

}  // namespace gs
