// rasterize_to_pixels forward / backward for gfx950 (the reference reaches
// it through /root/reference/3D-Reconstruction-main/models/trainers/base.py:393-408).
//
// MI355X-first decomposition: ONE WAVE64 PER 16x16 TILE, FOUR PIXELS (a 2x2 quad) PER LANE.
//   * a tile's sorted Gaussian list is staged through LDS in batches of 64 splat records (each lane
//     gathers one 64-B record = one HBM request, next batch prefetched into VGPRs behind the current one);
//   * every broadcast LDS read of a record is amortised over 4 pixels per lane (a 16x16-thread block would
//     issue 4x the LDS reads per pixel-Gaussian and be LDS-issue-bound on CDNA4's 256 B/clk LDS);
//   * a workgroup is a single wave: no cross-wave barrier anywhere in the inner loops, early termination
//     is one 64-bit ballot, and a tile's wave retires independently of its neighbours;
//   * backward: each lane first sums its 4 pixels' contributions in registers, then the wave does a
//     12-value reduce-scatter (v_permlane32_swap, v_permlane16_swap, DPP row mirrors) in 28 VALU ops and
//     12 lanes issue ONE atomic instruction covering one contiguous 48-B gradient record.
// No MFMA: nothing here is a dense contraction.  The forward is VALU/transcendental-issue bound, the
// backward VALU + atomic bound; HBM traffic is the 48-B-per-intersection gather + the image.
#include "gs_common.hpp"

namespace gs {

struct TileGeom {
  int cam, ty, tx;
  int start, end;
};

__device__ __forceinline__ TileGeom tile_geom(unsigned tile, int TW, int TH, unsigned n_tiles,
                                              const int32_t* __restrict__ offsets, int64_t n_isects) {
  TileGeom g;
  const int per_cam = TW * TH;
  g.cam = tile / per_cam;
  const int rem = tile - g.cam * per_cam;
  g.ty = rem / TW;
  g.tx = rem - g.ty * TW;
  g.start = offsets[tile];
  g.end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
  return g;
}

// ------------------------------------------------------------------------------------------ forward
// LDS image of one staged splat (12 floats), conic pre-scaled by -log2(e) so that vis = exp2(power):
//   q0 = {mx, my, A = -0.5*log2e*ca, B = -log2e*cb}   q1 = {C = -0.5*log2e*cc, opacity, c0, c1}   q2 = {c2, c3, -, -}
constexpr float LOG2E = 1.4426950408889634f;

__device__ __forceinline__ void stage_record(float4* sg, int lane, const float4& r0, const float4& r1, const float4& r2) {
  sg[lane * 3 + 0] = make_float4(r0.x, r0.y, -0.5f * LOG2E * r0.z, -LOG2E * r0.w);
  sg[lane * 3 + 1] = make_float4(-0.5f * LOG2E * r1.x, r1.y, r1.z, r1.w);
  sg[lane * 3 + 2] = r2;
}

// exponent (in log2 units) of the Gaussian at the lane's 2x2 pixel quad; dx/dy are mean - pixel centre
__device__ __forceinline__ void quad_power(const float4& q0, float Cc, float px0, float py0, float dx[2], float dy[2],
                                           float pw[4]) {
  dx[0] = q0.x - px0; dx[1] = dx[0] - 1.f;
  dy[0] = q0.y - py0; dy[1] = dy[0] - 1.f;
  const float ax0 = q0.z * dx[0], ax1 = q0.z * dx[1];
  const float cy0 = Cc * dy[0] * dy[0], cy1 = Cc * dy[1] * dy[1];
  pw[0] = fmaf(dx[0], fmaf(q0.w, dy[0], ax0), cy0);
  pw[1] = fmaf(dx[1], fmaf(q0.w, dy[0], ax1), cy0);
  pw[2] = fmaf(dx[0], fmaf(q0.w, dy[1], ax0), cy1);
  pw[3] = fmaf(dx[1], fmaf(q0.w, dy[1], ax1), cy1);
}

template <int CH>
__global__ __launch_bounds__(64) void rasterize_fwd_kernel(
    int C, int W, int H, int TW, int TH, const float* __restrict__ splats, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ flatten_ids, int64_t n_isects, const float* __restrict__ backgrounds, int ed,
    float* __restrict__ out_c, float* __restrict__ out_a, int32_t* __restrict__ last_ids) {
  __shared__ float4 sg[64 * 3];
  const unsigned n_tiles = (unsigned)(C * TW * TH);
  const unsigned tile = xcd_remap(blockIdx.x, n_tiles);
  const TileGeom tg = tile_geom(tile, TW, TH, n_tiles, offsets, n_isects);
  const int lane = threadIdx.x;
  const int ix0 = tg.tx * 16 + 2 * (lane & 7), iy0 = tg.ty * 16 + 2 * (lane >> 3);
  const float px0 = (float)ix0 + 0.5f, py0 = (float)iy0 + 0.5f;
  // T > 0: pixel live.  T < 0: pixel stopped (or outside the image); |T| is its final transmittance.
  float T[4], acc[4][CH];
  int cur[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const bool inside = (ix0 + (p & 1) < W) && (iy0 + (p >> 1) < H);
    T[p] = inside ? 1.f : -1.f;
    cur[p] = 0;
#pragma unroll
    for (int k = 0; k < CH; ++k) acc[p][k] = 0.f;
  }

  const int nbatch = (tg.end - tg.start + 63) >> 6;
  float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
  auto prefetch = [&](int b) {
    const int k = tg.start + b * 64 + lane;
    if (k < tg.end) {
      const float4* r = reinterpret_cast<const float4*>(splats + (int64_t)flatten_ids[k] * REC);
      n0 = r[0]; n1 = r[1]; n2 = r[2];
    }
  };
  if (nbatch > 0) prefetch(0);
  bool all_done = __ballot(T[0] > 0.f || T[1] > 0.f || T[2] > 0.f || T[3] > 0.f) == 0ull;
  for (int b = 0; b < nbatch && !all_done; ++b) {
    stage_record(sg, lane, n0, n1, n2);
    __syncthreads();
    if (b + 1 < nbatch) prefetch(b + 1);
    const int kbase = tg.start + b * 64;
    const int cnt = min(64, tg.end - kbase);
    for (int j = 0; j < cnt; ++j) {
      const float4 q0 = sg[j * 3 + 0];
      const float4 q1 = sg[j * 3 + 1];
      float dx[2], dy[2], pw[4], alpha[4];
      quad_power(q0, q1.x, px0, py0, dx, dy, pw);
      bool use[4];
      bool any_use = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        alpha[p] = fminf(ALPHA_MAX, q1.y * __builtin_amdgcn_exp2f(pw[p]));
        use[p] = (pw[p] <= 0.f) && (alpha[p] >= ALPHA_MIN) && (T[p] > 0.f);
        any_use |= use[p];
      }
      if (__ballot(any_use) == 0ull) continue;
      const float2 q2 = *reinterpret_cast<const float2*>(&sg[j * 3 + 2]);
      const float col[4] = {q1.z, q1.w, q2.x, q2.y};
      bool any_stop = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float nT = fmaf(-alpha[p], T[p], T[p]);
        const bool go = use[p] && (nT > T_MIN);
        const bool stop = use[p] && !(nT > T_MIN);
        any_stop |= stop;
        const float w = go ? alpha[p] * T[p] : 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) acc[p][k] = fmaf(col[k], w, acc[p][k]);
        cur[p] = go ? (kbase + j) : cur[p];
        T[p] = go ? nT : (stop ? -T[p] : T[p]);
      }
      if (__ballot(any_stop) != 0ull) {
        if (__ballot(T[0] > 0.f || T[1] > 0.f || T[2] > 0.f || T[3] > 0.f) == 0ull) { all_done = true; break; }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = ix0 + (p & 1), y = iy0 + (p >> 1);
    if (!(x < W && y < H)) continue;
    const int64_t pix = ((int64_t)tg.cam * H + y) * W + x;
    const float Tf = fabsf(T[p]);
    const float alpha_out = 1.f - Tf;
    out_a[pix] = alpha_out;
    last_ids[pix] = cur[p];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      float v = acc[p][k];
      if (backgrounds) v += Tf * backgrounds[tg.cam * CH + k];
      if (ed && k == CH - 1) v = v / fmaxf(alpha_out, 1e-10f);
      out_c[pix * CH + k] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ float dpp_add_row_mirror(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_row_half_mirror(float x) {
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x141, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_quad_xor1(float x) {  // quad_perm:[1,0,3,2]
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_add_quad_xor2(float x) {  // quad_perm:[2,3,0,1]
  return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x4E, 0xf, 0xf, false));
}
__device__ __forceinline__ float swap32_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Wave64 reduce-scatter of 12 per-lane values: on return every quad of lanes holds the complete
// 64-lane sum of ONE of the 12 inputs (or 0).  Which input a lane ends up with is fixed by the network
// and discovered once per wave by pushing index probes through it (see `slot` below).
__device__ __forceinline__ float reduce_scatter12(const float v[12], bool b3, bool b2) {
  float u[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) u[i] = swap32_add(v[2 * i], v[2 * i + 1]);
  float w[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) w[i] = swap16_add(u[2 * i], u[2 * i + 1]);
  const float t0 = dpp_add_row_mirror(w[0]);
  const float t1 = dpp_add_row_mirror(w[1]);
  const float t2 = dpp_add_row_mirror(w[2]);
  const float y0 = b3 ? t1 : t0;
  const float y1 = b3 ? 0.f : t2;
  const float z0 = dpp_add_row_half_mirror(y0);
  const float z1 = dpp_add_row_half_mirror(y1);
  float r = b2 ? z1 : z0;
  r = dpp_add_quad_xor2(r);
  r = dpp_add_quad_xor1(r);
  return r;
}

template <int CH, bool ABSGRAD>
__global__ __launch_bounds__(64) void rasterize_bwd_kernel(
    int C, int W, int H, int TW, int TH, const float* __restrict__ splats, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ flatten_ids, int64_t n_isects, const float* __restrict__ backgrounds, int ed,
    const float* __restrict__ out_c, const float* __restrict__ out_a, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_c, const float* __restrict__ v_out_a, float* __restrict__ v_splats) {
  __shared__ float4 sg[64 * 3];
  __shared__ int32_t sid[64];
  const unsigned n_tiles = (unsigned)(C * TW * TH);
  const unsigned tile = xcd_remap(blockIdx.x, n_tiles);
  const TileGeom tg = tile_geom(tile, TW, TH, n_tiles, offsets, n_isects);
  if (tg.end <= tg.start) return;
  const int lane = threadIdx.x;
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;

  // which of the 12 gradient slots does this lane hold after reduce_scatter12?  (self-calibrating)
  int slot;
  {
    float probe[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) probe[i] = (lane == 0) ? (float)(i + 1) : 0.f;
    slot = (int)reduce_scatter12(probe, b3, b2) - 1;
    if ((lane & 3) != 0) slot = -1;
  }

  const int ix0 = tg.tx * 16 + 2 * (lane & 7), iy0 = tg.ty * 16 + 2 * (lane >> 3);
  const float px[2] = {(float)ix0 + 0.5f, (float)ix0 + 1.5f};
  const float py[2] = {(float)iy0 + 0.5f, (float)iy0 + 1.5f};
  bool inside[4];
  float T[4], Tfa[4], buf[4][CH], vo[4][CH];
  float bgdot[4];
  int last[4];
  int my_max = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = ix0 + (p & 1), y = iy0 + (p >> 1);
    inside[p] = (x < W) && (y < H);
    T[p] = 1.f; Tfa[p] = 0.f; last[p] = -1; bgdot[p] = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) { buf[p][k] = 0.f; vo[p][k] = 0.f; }
    if (inside[p]) {
      const int64_t pix = ((int64_t)tg.cam * H + y) * W + x;
      const float A = out_a[pix];
      const float Tf = 1.f - A;
      float va = v_out_a[pix];
#pragma unroll
      for (int k = 0; k < CH; ++k) vo[p][k] = v_out_c[pix * CH + k];
      if (ed) {
        const float Ac = fmaxf(A, 1e-10f);
        const float ved = vo[p][CH - 1];
        const float edv = out_c[pix * CH + CH - 1];
        vo[p][CH - 1] = ved / Ac;
        if (A >= 1e-10f) va -= ved * edv / Ac;
      }
      if (backgrounds) {
        float accum = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) accum += backgrounds[tg.cam * CH + k] * vo[p][k];
        va -= accum;  // d(out)/dT * dT/dalpha folded: v_alpha gets -T_final*ra*accum
      }
      T[p] = Tf;
      Tfa[p] = Tf * va;
      last[p] = last_ids[pix];
      my_max = max(my_max, last[p]);
    }
  }
  (void)bgdot;
  // wave max of last ids
  int wmax = my_max;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
  const int k_top = min(tg.end - 1, wmax);

  float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
  int nid = 0;
  auto prefetch = [&](int kb) {
    const int k = kb - lane;
    if (k >= tg.start) {
      nid = flatten_ids[k];
      const float4* r = reinterpret_cast<const float4*>(splats + (int64_t)nid * REC);
      n0 = r[0]; n1 = r[1]; n2 = r[2];
    }
  };
  if (k_top >= tg.start) prefetch(k_top);
  for (int kb = k_top; kb >= tg.start; kb -= 64) {
    sg[lane * 3 + 0] = n0;
    sg[lane * 3 + 1] = n1;
    sg[lane * 3 + 2] = n2;
    sid[lane] = nid;
    __syncthreads();
    if (kb - 64 >= tg.start) prefetch(kb - 64);
    const int cnt = min(64, kb - tg.start + 1);
    for (int j = 0; j < cnt; ++j) {
      const int k = kb - j;
      const float4 a = sg[j * 3 + 0];   // mx my ca cb
      const float4 bq = sg[j * 3 + 1];  // cc op c0 c1
      const float4 cq = sg[j * 3 + 2];  // c2 c3 depth radius
      const float dx[2] = {a.x - px[0], a.x - px[1]};
      const float dy[2] = {a.y - py[0], a.y - py[1]};
      const float qx[2] = {a.z * dx[0] * dx[0], a.z * dx[1] * dx[1]};
      const float qy[2] = {bq.x * dy[0] * dy[0], bq.x * dy[1] * dy[1]};
      float alpha[4], vis[4];
      bool valid[4];
      bool any_valid = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float sigma = 0.5f * (qx[p & 1] + qy[p >> 1]) + a.w * dx[p & 1] * dy[p >> 1];
        vis[p] = __expf(-sigma);
        alpha[p] = fminf(ALPHA_MAX, bq.y * vis[p]);
        valid[p] = (k <= last[p]) && !(sigma < 0.f || alpha[p] < ALPHA_MIN);
        any_valid |= valid[p];
      }
      if (__ballot(any_valid) == 0ull) continue;
      const float col[4] = {bq.z, bq.w, cq.x, cq.y};
      float g[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) g[i] = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (valid[p]) {
          const float ra = 1.f / (1.f - alpha[p]);
          T[p] *= ra;
          const float fac = alpha[p] * T[p];
          float v_alpha = 0.f;
#pragma unroll
          for (int k2 = 0; k2 < CH; ++k2) {
            g[6 + k2] = fmaf(fac, vo[p][k2], g[6 + k2]);
            v_alpha += (col[k2] * T[p] - buf[p][k2] * ra) * vo[p][k2];
            buf[p][k2] = fmaf(col[k2], fac, buf[p][k2]);
          }
          v_alpha += Tfa[p] * ra;
          const float ov = bq.y * vis[p];
          if (ov <= ALPHA_MAX) {
            const float v_sigma = -ov * v_alpha;
            const float ddx = dx[p & 1], ddy = dy[p >> 1];
            g[2] = fmaf(0.5f * v_sigma * ddx, ddx, g[2]);
            g[3] = fmaf(v_sigma * ddx, ddy, g[3]);
            g[4] = fmaf(0.5f * v_sigma * ddy, ddy, g[4]);
            const float vx = v_sigma * (a.z * ddx + a.w * ddy);
            const float vy = v_sigma * (a.w * ddx + bq.x * ddy);
            g[0] += vx;
            g[1] += vy;
            if constexpr (ABSGRAD) {
              g[10] += fabsf(vx);
              g[11] += fabsf(vy);
            }
            g[5] = fmaf(vis[p], v_alpha, g[5]);
          }
        }
      }
      const float r = reduce_scatter12(g, b3, b2);
      if (slot >= 0) {
        if (ABSGRAD || slot < 10) atomicAdd(v_splats + (int64_t)sid[j] * REC + slot, r);
      }
    }
    __syncthreads();
  }
}

}  // namespace gs

static int raster_check(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                        int64_t n_isects) {
  if (C < 1 || width < 1 || height < 1 || n_isects < 0) return GS_ERR_INVALID_ARG;
  if (channels < 1 || channels > 4) return GS_ERR_UNSUPPORTED;
  if (tile_size != 16) return GS_ERR_UNSUPPORTED;
  if (n_isects >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  return GS_OK;
}

#define GS_RF_ARGS C, width, height, TW, TH, splats, isect_offsets, flatten_ids, n_isects, backgrounds, ed_last_channel, render_colors, render_alphas, last_ids
#define GS_RF_LAUNCH(CHN) hipLaunchKernelGGL((gs::rasterize_fwd_kernel<CHN>), grid, block, 0, s, GS_RF_ARGS)

extern "C" int gs_rasterize_fwd(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                                const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                int64_t n_isects, const float* backgrounds, int32_t ed_last_channel,
                                float* render_colors, float* render_alphas, int32_t* last_ids, void* stream) {
  int st = raster_check(C, width, height, tile_size, channels, n_isects);
  if (st != GS_OK) return st;
  if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return GS_ERR_INVALID_ARG;
  if (n_isects > 0 && (!splats || !flatten_ids)) return GS_ERR_INVALID_ARG;
  const int TW = (width + 15) / 16, TH = (height + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(C * TW * TH)), block(64);
  switch (channels) {
    case 1: GS_RF_LAUNCH(1); break;
    case 2: GS_RF_LAUNCH(2); break;
    case 3: GS_RF_LAUNCH(3); break;
    default: GS_RF_LAUNCH(4); break;
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

#define GS_RB_ARGS C, width, height, TW, TH, splats, isect_offsets, flatten_ids, n_isects, backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, v_render_colors, v_render_alphas, v_splats
#define GS_RB_LAUNCH(CHN)                                                                              \
  do {                                                                                                 \
    if (absgrad)                                                                                       \
      hipLaunchKernelGGL((gs::rasterize_bwd_kernel<CHN, true>), grid, block, 0, s, GS_RB_ARGS);        \
    else                                                                                               \
      hipLaunchKernelGGL((gs::rasterize_bwd_kernel<CHN, false>), grid, block, 0, s, GS_RB_ARGS);       \
  } while (0)

extern "C" int gs_rasterize_bwd(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                                const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                int64_t n_isects, const float* backgrounds, int32_t ed_last_channel,
                                const float* render_colors, const float* render_alphas, const int32_t* last_ids,
                                const float* v_render_colors, const float* v_render_alphas, int32_t absgrad,
                                float* v_splats, void* stream) {
  int st = raster_check(C, width, height, tile_size, channels, n_isects);
  if (st != GS_OK) return st;
  if (n_isects == 0) return GS_OK;
  if (!splats || !isect_offsets || !flatten_ids || !render_colors || !render_alphas || !last_ids ||
      !v_render_colors || !v_render_alphas || !v_splats)
    return GS_ERR_INVALID_ARG;
  const int TW = (width + 15) / 16, TH = (height + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)(C * TW * TH)), block(64);
  switch (channels) {
    case 1: GS_RB_LAUNCH(1); break;
    case 2: GS_RB_LAUNCH(2); break;
    case 3: GS_RB_LAUNCH(3); break;
    default: GS_RB_LAUNCH(4); break;
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}
