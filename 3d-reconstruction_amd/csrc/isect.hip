// isect_tiles (cumsum + emission), isect_offset_encode and the packed-mode compaction for gfx950.
// Integer work, bit-exact by construction.  HBM-bound.
#include "gs_common.hpp"

namespace gs {

// ------------------------------------------------------------------------------------------ scan
// Three-launch exclusive scan (reduce -> scan of block sums -> scan+add).  2048 items per workgroup:
// each lane owns 8 consecutive ints (two dwordx4 loads), wave64 DPP-free shuffle scan, LDS across waves.
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(v, o);
    if (lane >= o) v += t;
  }
  return v;
}

// block-wide exclusive scan of one value per thread; returns exclusive prefix, writes block total
__device__ __forceinline__ int64_t block_excl_scan(int64_t v, int64_t& total, int64_t* sh /*[4]*/) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int64_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int64_t t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) sh[wid] = incl;
  __syncthreads();
  int64_t base = 0;
  int64_t tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    if (w < wid) base += sh[w];
    tot += sh[w];
  }
  total = tot;
  __syncthreads();
  return base + incl - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                   int64_t* __restrict__ block_sums) {
  __shared__ int64_t sh[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i)
    if (base + i < n) s += in[base + i];
  int64_t total;
  block_excl_scan(s, total, sh);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single workgroup: exclusive scan of block_sums in place, total -> *total_out
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(int64_t* __restrict__ block_sums, int64_t nb,
                                                                       int64_t* __restrict__ total_out,
                                                                       int64_t tag = -1) {
  __shared__ int64_t sh[4];
  int64_t carry = 0;
  for (int64_t start = 0; start < nb; start += SCAN_THREADS) {
    const int64_t i = start + threadIdx.x;
    const int64_t v = (i < nb) ? block_sums[i] : 0;
    int64_t total;
    const int64_t ex = block_excl_scan(v, total, sh);
    if (i < nb) block_sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) {
    *total_out = carry;
    if (tag >= 0) total_out[1] = tag;  // gs_visibility_prefix: stat = {total, C}
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_final_kernel(const int32_t* __restrict__ in, int64_t n,
                                                                  const int64_t* __restrict__ block_sums,
                                                                  int32_t* __restrict__ out) {
  __shared__ int64_t sh[4];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
  int32_t v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    v[i] = (base + i < n) ? in[base + i] : 0;
    s += v[i];
  }
  int64_t total;
  int64_t ex = block_excl_scan(s, total, sh) + block_sums[blockIdx.x];
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; ++i) {
    if (base + i < n) out[base + i] = (int32_t)ex;
    ex += v[i];
  }
}

// ------------------------------------------------------------------------------------------ emit
// One lane per (camera, Gaussian).  Gaussians covering > 16 tiles are emitted cooperatively by the
// whole wave (contiguous 512-B key stores); small ones by their own lane.
__global__ __launch_bounds__(256) void isect_emit_kernel(int32_t N, int64_t n_slots, int32_t tile_w, int32_t tile_n_bits,
                                                         const float* __restrict__ splats,
                                                         const int32_t* __restrict__ tiles_per_gauss,
                                                         const int32_t* __restrict__ cum_excl,
                                                         int64_t* __restrict__ isect_ids,
                                                         int32_t* __restrict__ flatten_ids) {
  const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t idx = slot;  // flat row id c*N + n (A.3 emission order)
  int cnt = 0, x0 = 0, y0 = 0, wdt = 1;
  int64_t base = 0, key_hi = 0;
  if (slot < n_slots) {
    cnt = tiles_per_gauss[idx];
    if (cnt > 0) {
      const float4 r2 = reinterpret_cast<const float4*>(splats + idx * REC)[2];
      const float4 r3 = reinterpret_cast<const float4*>(splats + idx * REC)[3];
      const int mn = __float_as_int(r3.y), mx = __float_as_int(r3.z);
      x0 = mn & 0xffff; y0 = mn >> 16;
      wdt = (mx & 0xffff) - x0;
      base = cum_excl[slot];
      const int64_t cid = idx / N;
      key_hi = (cid << (32 + tile_n_bits)) | (int64_t)(uint32_t)__float_as_int(r2.z);
    }
  }
  constexpr int COOP = 16;
  if (cnt > 0 && cnt <= COOP) {
    for (int t = 0; t < cnt; ++t) {
      const int ti = y0 + t / wdt, tj = x0 + t % wdt;
      const int64_t tile_id = (int64_t)ti * tile_w + tj;
      isect_ids[base + t] = key_hi | (tile_id << 32);
      flatten_ids[base + t] = (int32_t)idx;
    }
  }
  const int lane = threadIdx.x & 63;
  unsigned long long big = __ballot(cnt > COOP);
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const int b_cnt = __shfl(cnt, src), b_x0 = __shfl(x0, src), b_y0 = __shfl(y0, src), b_w = __shfl(wdt, src);
    const int64_t b_base = __shfl(base, src), b_key = __shfl(key_hi, src);
    const int32_t b_idx = (int32_t)__shfl((int)idx, src);
    for (int t = lane; t < b_cnt; t += 64) {
      const int ti = b_y0 + t / b_w, tj = b_x0 + t % b_w;
      const int64_t tile_id = (int64_t)ti * tile_w + tj;
      isect_ids[b_base + t] = b_key | (tile_id << 32);
      flatten_ids[b_base + t] = b_idx;
    }
  }
}

// ------------------------------------------------------------------------------------------ offsets
__global__ __launch_bounds__(256) void offset_encode_kernel(int64_t n_isects, const int64_t* __restrict__ keys,
                                                            int32_t n_tiles_per_cam, int32_t tile_n_bits,
                                                            int64_t n_tiles_total, int32_t* __restrict__ offsets) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_isects) return;
  const int64_t tile_mask = ((int64_t)1 << tile_n_bits) - 1;
  auto lin = [&](int64_t key) {
    const int64_t hi = key >> 32;
    return (hi >> tile_n_bits) * n_tiles_per_cam + (hi & tile_mask);
  };
  const int64_t cur = lin(keys[k]);
  if (k == 0) {
    for (int64_t t = 0; t <= cur; ++t) offsets[t] = 0;
  } else {
    const int64_t prev = lin(keys[k - 1]);
    for (int64_t t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)k;
  }
  if (k == n_isects - 1) {
    for (int64_t t = cur + 1; t < n_tiles_total; ++t) offsets[t] = (int32_t)n_isects;
  }
}

// ------------------------------------------------------------------------------------------ packed mode
__global__ __launch_bounds__(256) void mark_visible_kernel(int64_t n, const int32_t* __restrict__ radii,
                                                           int32_t* __restrict__ visible) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) visible[i] = radii[i] > 0 ? 1 : 0;
}

__global__ __launch_bounds__(256) void compact_visible_kernel(
    int32_t C, int32_t N, const int32_t* __restrict__ radii, const int32_t* __restrict__ cum_excl,
    const float* __restrict__ splats, const float* __restrict__ means2d, int64_t* __restrict__ camera_ids,
    int64_t* __restrict__ gaussian_ids, int32_t* __restrict__ radii_p, float* __restrict__ means2d_p,
    float* __restrict__ depths_p, float* __restrict__ conics_p) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)C * N) return;
  const int r = radii[idx];
  if (r <= 0) return;
  const int64_t o = cum_excl[idx];
  const float4 r0 = reinterpret_cast<const float4*>(splats + idx * REC)[0];
  const float4 r1 = reinterpret_cast<const float4*>(splats + idx * REC)[1];
  const float4 r2 = reinterpret_cast<const float4*>(splats + idx * REC)[2];
  camera_ids[o] = idx / N;
  gaussian_ids[o] = idx % N;
  radii_p[o] = r;
  means2d_p[2 * o] = means2d[2 * idx];
  means2d_p[2 * o + 1] = means2d[2 * idx + 1];
  depths_p[o] = r2.z;
  conics_p[3 * o] = r0.z;
  conics_p[3 * o + 1] = r0.w;
  conics_p[3 * o + 2] = r1.x;
}

static inline int ilog2_floor(int64_t v) {
  int b = 0;
  while (v >>= 1) ++b;
  return b;
}

}  // namespace gs

extern "C" int64_t gs_exclusive_scan_workspace_bytes(int64_t n) {
  if (n < 0) return 0;
  return (gs::ceil_div(n, gs::SCAN_TILE) + 1) * (int64_t)sizeof(int64_t);
}

extern "C" int gs_exclusive_scan_i32(const int32_t* in, int32_t* out, int64_t n, int64_t* total, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
  if (n < 0 || !total) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    GS_HIP(hipMemsetAsync(total, 0, sizeof(int64_t), s));
    return GS_OK;
  }
  if (!in || !out || !workspace) return GS_ERR_INVALID_ARG;
  if (workspace_bytes < gs_exclusive_scan_workspace_bytes(n)) return GS_ERR_WORKSPACE;
  const int64_t nb = gs::ceil_div(n, gs::SCAN_TILE);
  int64_t* sums = (int64_t*)workspace;
  hipLaunchKernelGGL(gs::scan_reduce_kernel, dim3((unsigned)nb), dim3(gs::SCAN_THREADS), 0, s, in, n, sums);
  hipLaunchKernelGGL(gs::scan_block_sums_kernel, dim3(1), dim3(gs::SCAN_THREADS), 0, s, sums, nb, total,
                     (int64_t)-1);
  hipLaunchKernelGGL(gs::scan_final_kernel, dim3((unsigned)nb), dim3(gs::SCAN_THREADS), 0, s, in, n, sums, out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// Exclusive scan of the per-(camera, word) visible counts of gs_projection_fwd, and the two numbers that size the colour
// message: stat[0] = visible pairs, stat[1] = C.  (gs_exclusive_scan_i32's three launches; a single-workgroup,
// single-launch scan was measured at 59-155 us for the 78 k counts of configs[1] against 14 us for these three: one
// CU's memory pipeline is the limit there.)
extern "C" int gs_visibility_prefix(int32_t C, int32_t n_words, const int32_t* vis_counts, int32_t* vis_prefix,
                                    int64_t* stat, void* workspace, int64_t workspace_bytes, void* stream) {
  if (C < 1 || n_words < 0 || !stat) return GS_ERR_INVALID_ARG;
  const int64_t n = (int64_t)C * n_words;
  if (n >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (n == 0) {
    const int64_t host[2] = {0, C};
    GS_HIP(hipMemcpyAsync(stat, host, sizeof(host), hipMemcpyHostToDevice, s));
    return GS_OK;
  }
  if (!vis_counts || !vis_prefix || !workspace) return GS_ERR_INVALID_ARG;
  if (workspace_bytes < gs_exclusive_scan_workspace_bytes(n)) return GS_ERR_WORKSPACE;
  const int64_t nb = gs::ceil_div(n, gs::SCAN_TILE);
  int64_t* sums = (int64_t*)workspace;
  hipLaunchKernelGGL(gs::scan_reduce_kernel, dim3((unsigned)nb), dim3(gs::SCAN_THREADS), 0, s, vis_counts, n, sums);
  hipLaunchKernelGGL(gs::scan_block_sums_kernel, dim3(1), dim3(gs::SCAN_THREADS), 0, s, sums, nb, stat, (int64_t)C);
  hipLaunchKernelGGL(gs::scan_final_kernel, dim3((unsigned)nb), dim3(gs::SCAN_THREADS), 0, s, vis_counts, n, sums,
                     vis_prefix);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

static int emit_check(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int* tile_n_bits) {
  if (C < 1 || N < 0 || tile_w < 1 || tile_h < 1) return GS_ERR_INVALID_ARG;
  *tile_n_bits = gs::ilog2_floor((int64_t)tile_w * tile_h) + 1;
  const int cam_n_bits = gs::ilog2_floor(C) + 1;
  if (32 + *tile_n_bits + cam_n_bits > 64) return GS_ERR_UNSUPPORTED;
  if ((int64_t)C * N >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  return GS_OK;
}

extern "C" int gs_isect_emit(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                             const int32_t* tiles_per_gauss, const int32_t* cum_excl, int64_t* isect_ids,
                             int32_t* flatten_ids, void* stream) {
  int tile_n_bits;
  int st = emit_check(C, N, tile_w, tile_h, &tile_n_bits);
  if (st != GS_OK) return st;
  if (N == 0) return GS_OK;
  if (!splats || !tiles_per_gauss || !cum_excl || !isect_ids || !flatten_ids) return GS_ERR_INVALID_ARG;
  const int64_t total = (int64_t)C * N;
  hipLaunchKernelGGL(gs::isect_emit_kernel, dim3((unsigned)gs::ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, N, total, tile_w, tile_n_bits, splats, tiles_per_gauss, cum_excl, isect_ids,
                     flatten_ids);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_offset_encode(int64_t n_isects, const int64_t* isect_ids_sorted, int32_t C, int32_t tile_w,
                                      int32_t tile_h, int32_t* offsets, void* stream) {
  if (n_isects < 0 || C < 1 || tile_w < 1 || tile_h < 1 || !offsets) return GS_ERR_INVALID_ARG;
  if (n_isects >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_tiles = (int64_t)C * tile_w * tile_h;
  if (n_isects == 0) {
    GS_HIP(hipMemsetAsync(offsets, 0, n_tiles * sizeof(int32_t), s));
    return GS_OK;
  }
  if (!isect_ids_sorted) return GS_ERR_INVALID_ARG;
  const int tile_n_bits = gs::ilog2_floor((int64_t)tile_w * tile_h) + 1;
  hipLaunchKernelGGL(gs::offset_encode_kernel, dim3((unsigned)gs::ceil_div(n_isects, 256)), dim3(256), 0, s,
                     n_isects, isect_ids_sorted, tile_w * tile_h, tile_n_bits, n_tiles, offsets);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_mark_visible(int64_t n, const int32_t* radii, int32_t* visible, void* stream) {
  if (n < 0) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!radii || !visible) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::mark_visible_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, radii, visible);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_compact_visible(int32_t C, int32_t N, const int32_t* radii, const int32_t* cum_excl,
                                  const float* splats, const float* means2d, int64_t* camera_ids,
                                  int64_t* gaussian_ids, int32_t* radii_packed, float* means2d_packed,
                                  float* depths_packed, float* conics_packed, void* stream) {
  if (C < 1 || N < 0) return GS_ERR_INVALID_ARG;
  if (N == 0) return GS_OK;
  if (!radii || !cum_excl || !splats || !means2d || !camera_ids || !gaussian_ids || !radii_packed ||
      !means2d_packed || !depths_packed || !conics_packed)
    return GS_ERR_INVALID_ARG;
  const int64_t total = (int64_t)C * N;
  hipLaunchKernelGGL(gs::compact_visible_kernel, dim3((unsigned)gs::ceil_div(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, C, N, radii, cum_excl, splats, means2d, camera_ids, gaussian_ids,
                     radii_packed, means2d_packed, depths_packed, conics_packed);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
