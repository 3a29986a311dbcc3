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

// ---- contract flag (api.hip) ------------------------------------------------------------------------------------
// A kernel that finds its inputs violating an entry point's documented precondition (not a bug it can compute through)
// raises this process-wide flag -- one int32 in pinned, coherent host memory, written with a system-scope store -- and
// leaves its outputs in a state no later kernel faults on.  Host side: the entry points of the same family look at the
// flag when they are called (a plain host load, no synchronisation) and return GS_ERR_CONTRACT once, clearing it;
// gs_contract_status() reads it on demand.  NULL from gs_contract_flag_device() (allocation failed) = no reporting.
// (library-internal: not part of the C ABI)
__attribute__((visibility("hidden"))) int32_t* gs_contract_flag_device();   // device-visible address for kernel arguments
__attribute__((visibility("hidden"))) int gs_contract_take();               // host: current code (0 = none), cleared by the read

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

// ---- wave-cooperative access to [n, 3] fp32 arrays ------------------------------------------------------------------
// A lane reading (or writing) its own 12-B row makes the wave issue three dword (or one unaligned dwordx3) accesses at a
// 12-B lane stride; measured on gfx950 (scratch/ubench/readbw.hip) one such read plus one such write per row cost a
// 192-B-per-row stream a quarter of its rate.  The wave's 64 rows are 768 contiguous bytes: move them as 48 aligned
// float4 and transpose through a wave-private LDS tile of 192 floats (row stride 3 dwords: conflict-free).
// `row0` = the wave's first row (a multiple of 64), `rows` = how many of its 64 rows exist; base 16-B aligned.
__device__ __forceinline__ void wave_load3(const float* __restrict__ base, int64_t row0, int rows, int lane,
                                           float* __restrict__ tile, float out[3]) {
  const int total = rows * 3;
  const float* src = base + row0 * 3;
  if (4 * lane + 3 < total) {
    *reinterpret_cast<float4*>(tile + 4 * lane) = *reinterpret_cast<const float4*>(src + 4 * lane);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * lane + k < total) tile[4 * lane + k] = src[4 * lane + k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int r = lane < rows ? lane : (rows > 0 ? rows - 1 : 0);
  out[0] = tile[3 * r]; out[1] = tile[3 * r + 1]; out[2] = tile[3 * r + 2];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void wave_store3(float* __restrict__ base, int64_t row0, int rows, int lane,
                                            float* __restrict__ tile, const float v[3]) {
  tile[3 * lane] = v[0]; tile[3 * lane + 1] = v[1]; tile[3 * lane + 2] = v[2];
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int total = rows * 3;
  float* dst = base + row0 * 3;
  if (4 * lane + 3 < total) {
    *reinterpret_cast<float4*>(dst + 4 * lane) = *reinterpret_cast<const float4*>(tile + 4 * lane);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (4 * lane + k < total) dst[4 * lane + k] = tile[4 * lane + k];
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// row `n` of a [*, 3] array: cooperatively when the array is 16-B aligned (`coop`, wave-uniform), else lane by lane
__device__ __forceinline__ void load3(const float* __restrict__ base, bool coop, int n, int64_t row0, int rows, int lane,
                                      float* __restrict__ tile, float out[3]) {
  if (coop) {
    wave_load3(base, row0, rows, lane, tile, out);
  } else {
    out[0] = base[3 * (int64_t)n]; out[1] = base[3 * (int64_t)n + 1]; out[2] = base[3 * (int64_t)n + 2];
  }
}
__device__ __forceinline__ void store3(float* __restrict__ base, bool coop, int n, bool live, int64_t row0, int rows,
                                       int lane, float* __restrict__ tile, const float v[3]) {
  if (coop) {
    wave_store3(base, row0, rows, lane, tile, v);
  } else if (live) {
    base[3 * (int64_t)n] = v[0]; base[3 * (int64_t)n + 1] = v[1]; base[3 * (int64_t)n + 2] = v[2];
  }
}

}  // namespace gs
