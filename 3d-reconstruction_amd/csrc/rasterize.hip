// rasterize_to_pixels forward / backward for gfx950 (the reference reaches
// it through /root/reference/3D-Reconstruction-main/models/trainers/base.py:393-408).
//
// MI355X-first decomposition: ONE WAVE64 PER 16x16 TILE, FOUR PIXELS (a 2x2 quad) PER LANE, and the wave's four
// DPP rows (16 lanes each) own the tile's four 8x8 QUADRANTS.
//   * a tile's sorted Gaussian list is staged through LDS in batches of 64 splat records (each lane
//     gathers one 64-B record = one HBM request, next batch prefetched into VGPRs behind the current one);
//   * at staging ONE lane per splat culls it exactly against each quadrant, and every quadrant gets its own
//     order-preserving queue of LDS record offsets: in one inner-loop iteration the four rows composite FOUR
//     DIFFERENT splats, so a splat that touches one corner of the tile costs a quarter of a wave-iteration, and
//     a quadrant whose 64 pixels have all terminated stops taking splats (early termination per 8x8 block);
//   * every LDS read of a record is amortised over 4 pixels per lane (a 16x16-thread block would
//     issue 4x the LDS reads per pixel-Gaussian and be LDS-issue-bound on CDNA4's 256 B/clk LDS);
//   * a workgroup is a single wave: no cross-wave barrier anywhere in the inner loops, early termination
//     is one 64-bit ballot, and a tile's wave retires independently of its neighbours;
//   * backward: one splat per wave-iteration (whole-tile cull).  Each lane first sums its 4 pixels' contributions
//     in registers, then the wave does a 12-value reduce-scatter (v_permlane32_swap, v_permlane16_swap, DPP row
//     mirrors) in 28 VALU ops and 12 lanes issue ONE atomic instruction covering one contiguous 48-B gradient
//     record.  Quadrant queues were measured here too and lose: four rows adding the SAME splat's record within a
//     microsecond of each other serialise at the memory-side atomic units (0.47 -> 0.71 ms), and merging them in
//     LDS first (ds_add_f32 + a per-batch flush) costs what the shorter walk saves (0.49 ms).
// No MFMA: nothing here is a dense contraction.  The forward is VALU/transcendental-issue bound, the
// backward VALU + atomic bound; HBM traffic is the 48-B-per-intersection gather + the image.
#include "gs_common.hpp"
#include "raster_cull.hpp"

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

// Workgroup -> tile map.  Workgroups b and b+8 share an XCD (observed round-robin dispatch), so runs of
// RASTER_CHUNK consecutive tiles (neighbours gather the same splat records -> same 4 MiB L2) go to one XCD
// and successive chunks rotate over the 8 XCDs, which keeps cheap regions (sky, a sparse camera) from
// idling whole XCDs.  The grid is padded to a multiple of 8 * RASTER_CHUNK; surplus workgroups exit.
constexpr unsigned RASTER_CHUNK = 16;
__host__ __device__ inline unsigned raster_grid(unsigned n_tiles) {
  const unsigned n_chunks = (n_tiles + RASTER_CHUNK - 1) / RASTER_CHUNK;
  return ((n_chunks + 7u) / 8u) * 8u * RASTER_CHUNK;
}
__device__ __forceinline__ unsigned raster_tile_of(unsigned bid) {
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  return ((idx / RASTER_CHUNK) * 8u + xcd) * RASTER_CHUNK + (idx % RASTER_CHUNK);
}

// LDS image of one staged splat, conic pre-scaled so that vis = exp2(pw) with
//   pw(dx, dy) = A dx^2 + 2 Bh dx dy + C dy^2,   (A, Bh, C) = -0.5 * log2(e) * (conic a, b, c),  d = mean - pixel
//   q0 = {mx, my, A, Bh}   q1 = {C, opacity, c0, c1}   q2 = {c2, c3, sorted index k | gaussian row, -}
// ------------------------------------------------------------------------------------------ quadrant queues
// LDS image of one staged batch: 64 record slots + one NULL record (all zero: opacity 0 -> alpha 0, never used by a
// pixel) and, per quadrant, a queue of u16 byte offsets into the record area in list order, padded with the NULL
// record's offset up to the longest queue of the batch (the inner loop reads one entry ahead, hence the fixed tail).
constexpr int Q_SLOTS = 65;
constexpr int Q_REC_BYTES = 48;
constexpr int Q_NULL_OFF = 64 * Q_REC_BYTES;
constexpr int Q_STRIDE = 68;  // u16 entries per queue: 64 + a NULL tail

__device__ __forceinline__ void stage_init(float4* __restrict__ sg, unsigned short* __restrict__ qidx, int lane) {
  // {mean 0, conic 0} {C 0, opacity 0, colour 0} {colour 0, list index 0, gradient row -1 (bits, never a float)}
  if (lane < 3) sg[64 * 3 + lane] = make_float4(0.f, 0.f, 0.f, lane == 2 ? __int_as_float(-1) : 0.f);
  if (lane < 16) qidx[(lane >> 2) * Q_STRIDE + 64 + (lane & 3)] = (unsigned short)Q_NULL_OFF;
}

// One lane per staged list entry: scale the conic, cull the splat exactly against each quadrant it is offered to
// (`qmask` bit q; 0 for lanes past the end of the list), write the record, append its offset to the queues of the
// quadrants it may touch (ballot compaction, list order preserved).  Returns the longest queue's length.
__device__ __forceinline__ int stage_batch(float4* __restrict__ sg, unsigned short* __restrict__ qidx, int lane,
                                           unsigned qmask, float4 n0, float4 n1, float4 rec2, float x0, float y0) {
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float A = CONIC_SCALE * n0.z, Bh = CONIC_SCALE * n0.w, Cc = CONIC_SCALE * n1.x;
  int cnt[4];
  int nmax = 0;
  bool any = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool m = ((qmask >> q) & 1u) &&
                   rect_may_touch(n0.x, n0.y, A, Bh, Cc, n1.y, x0 + 8.f * (float)(q & 1), y0 + 8.f * (float)(q >> 1), 7.f);
    const unsigned long long bal = __ballot(m);
    cnt[q] = __popcll(bal);
    if (m) qidx[q * Q_STRIDE + __popcll(bal & lt_mask)] = (unsigned short)(lane * Q_REC_BYTES);
    any |= m;
    nmax = max(nmax, cnt[q]);
  }
  if (any) {
    sg[lane * 3 + 0] = make_float4(n0.x, n0.y, A, Bh);
    sg[lane * 3 + 1] = make_float4(Cc, n1.y, n1.z, n1.w);
    sg[lane * 3 + 2] = rec2;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (lane >= cnt[q] && lane <= nmax) qidx[q * Q_STRIDE + lane] = (unsigned short)Q_NULL_OFF;
  return nmax;
}

__device__ __forceinline__ unsigned quadrants_of(unsigned long long lanes) {  // bit q: some lane of row q is set
  return ((lanes & 0xFFFFull) ? 1u : 0u) | ((lanes & 0xFFFF0000ull) ? 2u : 0u) |
         ((lanes & 0xFFFF00000000ull) ? 4u : 0u) | ((lanes & 0xFFFF000000000000ull) ? 8u : 0u);
}

// ------------------------------------------------------------------------------------------ forward
template <int CH>
__global__ __launch_bounds__(64) void rasterize_fwd_kernel(
    int C, int W, int H, int TW, int TH, const float* __restrict__ splats, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ flatten_ids, int64_t n_isects, const float* __restrict__ backgrounds, int ed,
    float* __restrict__ out_c, float* __restrict__ out_a, int32_t* __restrict__ last_ids,
    const int32_t* __restrict__ tile_order, int32_t* __restrict__ tile_work,
    const int32_t* __restrict__ sorted_cnt, int32_t* __restrict__ redo_flags, int redo_pass,
    const float* __restrict__ opac_override, int n_per_cam, float* __restrict__ out_T,
    const int64_t* __restrict__ n_dev) {
  __shared__ float4 sg[Q_SLOTS * 3];
  // n_dev: the intersection count in device memory, `n_isects` the capacity the lists were allocated with before the
  // host knew the count (gs_rasterize_fwd_dev); a count beyond it means the lists were never built: write nothing
  if (n_dev) {
    const int64_t n_true = *n_dev;
    if (n_true > n_isects) return;
    n_isects = n_true;
  }
  __shared__ unsigned short qidx[4 * Q_STRIDE];
  const unsigned n_tiles = (unsigned)(C * TW * TH);
  unsigned tile;
  if (tile_order) {
    if (blockIdx.x >= n_tiles) return;
    tile = (unsigned)tile_order[blockIdx.x];
  } else {
    tile = raster_tile_of(blockIdx.x);
    if (tile >= n_tiles) return;
  }
  // second pass: only the tiles whose sorted front ran out (redo_flags[n_tiles] counts them: usually none)
  if (redo_pass && (redo_flags[n_tiles] == 0 || redo_flags[tile] == 0)) return;
  TileGeom tg = tile_geom(tile, TW, TH, n_tiles, offsets, n_isects);
  // front-sorted lists (gs_isect_tile_sort_front): only the first sorted_cnt[tile] entries are in order yet
  bool truncated = false;
  if (sorted_cnt && !redo_pass) {
    const int sc = sorted_cnt[tile];
    if (sc < tg.end - tg.start) {
      tg.end = tg.start + max(sc, 0);
      truncated = true;
    }
  }
  const int lane = threadIdx.x;
  const int quad = lane >> 4, rl = lane & 15;  // DPP row = quadrant; 4x4 lanes of 2x2 pixels inside it
  const float x0 = (float)(tg.tx * 16) + 0.5f, y0 = (float)(tg.ty * 16) + 0.5f;
  const int ix0 = tg.tx * 16 + 8 * (quad & 1) + 2 * (rl & 3), iy0 = tg.ty * 16 + 8 * (quad >> 1) + 2 * (rl >> 2);
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
  stage_init(sg, qidx, lane);

  const int nbatch = (tg.end - tg.start + 63) >> 6;
  float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
  auto prefetch = [&](int b) {
    const int k = tg.start + b * 64 + lane;
    if (k < tg.end) {
      const int row = flatten_ids[k];
      const float4* r = reinterpret_cast<const float4*>(splats + (int64_t)row * REC);
      n0 = r[0]; n1 = r[1]; n2 = r[2];
      // re-render of cached records with other opacities (per-class masks of the eval path): opacity x compensation
      if (opac_override) n1.y = opac_override[row % n_per_cam] * splats[(int64_t)row * REC + 12];
    }
  };
  if (nbatch > 0) prefetch(0);
  unsigned long long live = __ballot(T[0] > 0.f || T[1] > 0.f || T[2] > 0.f || T[3] > 0.f);
  const unsigned short* myq = qidx + quad * Q_STRIDE;
  for (int b = 0; b < nbatch && live != 0ull; ++b) {
    const int kbase = tg.start + b * 64;
    // a quadrant whose pixels have all stopped takes no more splats
    const unsigned qmask = (kbase + lane < tg.end) ? quadrants_of(live) : 0u;
    const int nmax = stage_batch(sg, qidx, lane, qmask, n0, n1,
                                 make_float4(n2.x, n2.y, __int_as_float(kbase + lane), 0.f), x0, y0);
    __syncthreads();
    if (b + 1 < nbatch) prefetch(b + 1);
    unsigned off = myq[0];
    for (int j = 0; j < nmax; ++j) {
      const float4* rec = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sg) + off);
      off = myq[j + 1];
      const float4 q0 = rec[0];
      const float4 q1 = rec[1];
      const float dx0 = q0.x - px0, dx1 = dx0 - 1.f;
      const float dy0 = q0.y - py0, dy1 = dy0 - 1.f;
      const float adx[2] = {q0.z * dx0, q0.z * dx1}, bdx[2] = {q0.w * dx0, q0.w * dx1};
      const float bdy[2] = {q0.w * dy0, q0.w * dy1}, cdy[2] = {q1.x * dy0, q1.x * dy1};
      const float dxv[2] = {dx0, dx1}, dyv[2] = {dy0, dy1};
      float pw[4], alpha[4];
      bool use[4];
      bool any_use = false;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int i = p & 1, jy = p >> 1;
        pw[p] = fmaf(dxv[i], adx[i] + bdy[jy], dyv[jy] * (bdx[i] + cdy[jy]));
        alpha[p] = fminf(ALPHA_MAX, q1.y * __builtin_amdgcn_exp2f(pw[p]));
        use[p] = (pw[p] <= 0.f) && (alpha[p] >= ALPHA_MIN) && (T[p] > 0.f);
        any_use |= use[p];
      }
      if (__ballot(any_use) == 0ull) continue;
      const float4 q2 = rec[2];
      const float col[4] = {q1.z, q1.w, q2.x, q2.y};
      const int kcur = __float_as_int(q2.z);
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
        cur[p] = go ? kcur : cur[p];
        T[p] = go ? nT : (stop ? -T[p] : T[p]);
      }
      if (__ballot(any_stop) != 0ull) {
        live = __ballot(T[0] > 0.f || T[1] > 0.f || T[2] > 0.f || T[3] > 0.f);
        if (live == 0ull) break;
      }
    }
    __syncthreads();
  }

  // pixels still live where the sorted front ends: the tile needs its whole list (sorted by gs_isect_tile_sort_redo)
  // and is rasterized again by the redo pass, which overwrites everything written below
  if (truncated && live != 0ull && lane == 0) {
    redo_flags[tile] = 1;
    atomicAdd(&redo_flags[n_tiles], 1);
  }
  if (tile_work) {
    // list entries the backward will walk for this tile (it starts at the wave's largest last id)
    int wmax = max(max(cur[0], cur[1]), max(cur[2], cur[3]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    if (lane == 0) tile_work[tile] = (tg.end > tg.start) ? max(wmax - tg.start + 1, 0) : 0;
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = ix0 + (p & 1), y = iy0 + (p >> 1);
    if (!(x < W && y < H)) continue;
    const int64_t pix = ((int64_t)tg.cam * H + y) * W + x;
    const float Tf = fabsf(T[p]);
    const float alpha_out = 1.f - Tf;
    out_a[pix] = alpha_out;
    if (out_T) out_T[pix] = Tf;  // the backward's starting point, exact (1 - alpha_out has lost Tf's low bits)
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

// Backward (A.6), branch-free over the lane's 4 pixels.  Per pixel the running state is two scalars:
//   T  transmittance in front of the splat being visited (T_final divided back by (1 - alpha) one splat at a time)
//   E  = T_final * v_alpha_out' - sum_{splats behind} fac * <colour, v_out>     (so that
//        v_alpha = T * <colour, v_out> + E / (1 - alpha); this replaces A.6's per-channel `buf` accumulator)
// A pixel for which the splat is not valid (k > last id, sigma < 0, alpha < 1/255) gets alpha := 0, which turns
// every update into an exact no-op (ra = 1, fac = 0, z = 0) without a branch.  Gradient sums are accumulated in
// the scaled conic units and rescaled once per slot after the wave reduction (POST below).
template <int CH, bool ABSGRAD>
__global__ __launch_bounds__(64) void rasterize_bwd_kernel(
    int C, int W, int H, int TW, int TH, const float* __restrict__ splats, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ flatten_ids, int64_t n_isects, const float* __restrict__ backgrounds, int ed,
    const float* __restrict__ out_c, const float* __restrict__ out_a, const int32_t* __restrict__ last_ids,
    const float* __restrict__ v_out_c, const float* __restrict__ v_out_a, float* __restrict__ v_splats,
    const int32_t* __restrict__ tile_order, const float* __restrict__ final_T) {
  __shared__ float4 sg[64 * 3];
  const unsigned n_tiles = (unsigned)(C * TW * TH);
  unsigned tile;
  if (tile_order) {
    if (blockIdx.x >= n_tiles) return;
    tile = (unsigned)tile_order[blockIdx.x];
  } else {
    tile = raster_tile_of(blockIdx.x);
    if (tile >= n_tiles) return;
  }
  const TileGeom tg = tile_geom(tile, TW, TH, n_tiles, offsets, n_isects);
  if (tg.end <= tg.start) return;
  const int lane = threadIdx.x;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0;

  // which of the 12 gradient slots does this lane hold after reduce_scatter12?  (self-calibrating)
  int slot;
  {
    float probe[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) probe[i] = (lane == 0) ? (float)(i + 1) : 0.f;
    slot = (int)reduce_scatter12(probe, b3, b2) - 1;
    if ((lane & 3) != 0) slot = -1;
    if (!ABSGRAD && slot >= 10) slot = -1;
  }
  // slot -> true-gradient scale: mean (0,1) and absgrad (10,11): 2 ln2; conic (2,3,4): -1/2, -1, -1/2; rest 1
  float post = 1.f;
  if (slot == 0 || slot == 1 || slot >= 10) post = 2.f * LN2;
  if (slot == 2 || slot == 4) post = -0.5f;
  if (slot == 3) post = -1.f;

  // a lane owns the same pixel of each of the tile's four 8x8 QUADRANTS (p = quadrant): a splat that reaches only some
  // quadrants -- 2.7 of 4 on average -- skips the others' pixel work on a wave-uniform branch, where a 2x2 quad per
  // lane would evaluate all 256 pixels for every splat
  const float x0 = (float)(tg.tx * 16) + 0.5f, y0 = (float)(tg.ty * 16) + 0.5f;
  const int ix0 = tg.tx * 16 + (lane & 7), iy0 = tg.ty * 16 + (lane >> 3);
  const float px0 = (float)ix0 + 0.5f, py0 = (float)iy0 + 0.5f;
  float T[4], E[4], vo[4][CH];
  int last[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = ix0 + 8 * (p & 1), y = iy0 + 8 * (p >> 1);
    T[p] = 1.f; E[p] = 0.f; last[p] = -1;
#pragma unroll
    for (int k = 0; k < CH; ++k) vo[p][k] = 0.f;
    if ((x < W) && (y < H)) {
      const int64_t pix = ((int64_t)tg.cam * H + y) * W + x;
      const float Aout = out_a[pix];
      // gsplat recovers T_final as 1 - render_alphas, which in fp32 carries an absolute error of half an ulp of
      // alpha ~ 3e-8, i.e. a RELATIVE error of up to 3e-4 on the T_final ~ 1e-4 of a saturated pixel, and every
      // gradient term of that pixel scales with it.  The forward's own T_final removes that term.
      const float Tf = final_T ? final_T[pix] : 1.f - Aout;
      float va = v_out_a[pix];
#pragma unroll
      for (int k = 0; k < CH; ++k) vo[p][k] = v_out_c[pix * CH + k];
      if (ed) {
        const float Ac = fmaxf(Aout, 1e-10f);
        const float ved = vo[p][CH - 1];
        const float edv = out_c[pix * CH + CH - 1];
        vo[p][CH - 1] = ved / Ac;
        if (Aout >= 1e-10f) va -= ved * edv / Ac;
      }
      if (backgrounds) {
        float accum = 0.f;
#pragma unroll
        for (int k = 0; k < CH; ++k) accum += backgrounds[tg.cam * CH + k] * vo[p][k];
        va -= accum;  // d(out)/dT * dT/dalpha folded: v_alpha gets -T_final*ra*accum
      }
      T[p] = Tf;
      E[p] = Tf * va;
      last[p] = last_ids[pix];
    }
  }
  // wave max of last ids, and per quadrant: entries behind a quadrant's last contributor skip it without evaluation
  int qlast[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    int m = last[p];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    qlast[p] = __builtin_amdgcn_readfirstlane(m);
  }
  const int wmax = max(max(qlast[0], qlast[1]), max(qlast[2], qlast[3]));
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
    const float A = CONIC_SCALE * n0.z, Bh = CONIC_SCALE * n0.w, Cc = CONIC_SCALE * n1.x;
    // exact cull against each quadrant (one lane per staged splat): the quadrants it may reach, as a 4-bit mask
    unsigned qm = 0u;
    if (kb - lane >= tg.start) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (rect_may_touch(n0.x, n0.y, A, Bh, Cc, n1.y, x0 + 8.f * (float)(q & 1), y0 + 8.f * (float)(q >> 1), 7.f))
          qm |= 1u << q;
    }
    const bool keep = qm != 0u;
    const unsigned long long kmask = __ballot(keep);
    const int cnt = __popcll(kmask);
    if (keep) {
      const int pos = __popcll(kmask & lt_mask);
      sg[pos * 3 + 0] = make_float4(n0.x, n0.y, A, Bh);
      sg[pos * 3 + 1] = make_float4(Cc, n1.y, n1.z, n1.w);
      // (list index as its distance below kb: 6 bits; the quadrant mask above it)
      sg[pos * 3 + 2] = make_float4(n2.x, n2.y, __int_as_float(lane | (int)(qm << 8)), __int_as_float(nid));
    }
    __syncthreads();
    if (kb - 64 >= tg.start) prefetch(kb - 64);
    for (int j = 0; j < cnt; ++j) {
      const float4 q0 = sg[j * 3 + 0];  // mx my A Bh
      const float4 q1 = sg[j * 3 + 1];  // C op c0 c1
      const float4 q2 = sg[j * 3 + 2];  // c2 c3 (k offset | quadrant mask) gid
      const int meta = __builtin_amdgcn_readfirstlane(__float_as_int(q2.z));  // (every lane read the same slot)
      const int k = kb - (meta & 63);
      const unsigned quads = (unsigned)meta >> 8;
      const float dx0 = q0.x - px0, dx1 = dx0 - 8.f;
      const float dy0 = q0.y - py0, dy1 = dy0 - 8.f;
      const float adx[2] = {q0.z * dx0, q0.z * dx1}, bdx[2] = {q0.w * dx0, q0.w * dx1};
      const float bdy[2] = {q0.w * dy0, q0.w * dy1}, cdy[2] = {q1.x * dy0, q1.x * dy1};
      const float dxv[2] = {dx0, dx1}, dyv[2] = {dy0, dy1};
      const float col[4] = {q1.z, q1.w, q2.x, q2.y};
      float g[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) g[i] = 0.f;
      bool contributed = false;  // wave-uniform
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        // wave-uniform: no pixel of this quadrant can take the splat, or all of them stopped in front of it
        if (!((quads >> p) & 1u) || k > qlast[p]) continue;
        const int i = p & 1, jy = p >> 1;
        const float gx = adx[i] + bdy[jy];
        const float gy = bdx[i] + cdy[jy];
        const float pw = fmaf(dxv[i], gx, dyv[jy] * gy);
        const float vis = __builtin_amdgcn_exp2f(pw);
        const float ov = q1.y * vis;
        const float alpha = fminf(ALPHA_MAX, ov);
        const bool valid = (k <= last[p]) && (pw <= 0.f) && (alpha >= ALPHA_MIN);
        if (__ballot(valid) == 0ull) continue;
        contributed = true;
        const float ae = valid ? alpha : 0.f;
        const float ra = __builtin_amdgcn_rcpf(1.f - ae);
        T[p] *= ra;
        const float fac = ae * T[p];
        float cdot = col[0] * vo[p][0];
#pragma unroll
        for (int c = 1; c < CH; ++c) cdot = fmaf(col[c], vo[p][c], cdot);
        const float v_alpha = fmaf(T[p], cdot, ra * E[p]);
        E[p] = fmaf(-fac, cdot, E[p]);
#pragma unroll
        for (int c = 0; c < CH; ++c) g[6 + c] = fmaf(fac, vo[p][c], g[6 + c]);
        const float vag = (valid && ov <= ALPHA_MAX) ? v_alpha : 0.f;
        g[5] = fmaf(vis, vag, g[5]);
        const float z = ov * vag;  // = -v_sigma
        const float zdx = z * dxv[i], zdy = z * dyv[jy];
        g[2] = fmaf(zdx, dxv[i], g[2]);
        g[3] = fmaf(zdx, dyv[jy], g[3]);
        g[4] = fmaf(zdy, dyv[jy], g[4]);
        const float vx = z * gx, vy = z * gy;
        g[0] += vx;
        g[1] += vy;
        if constexpr (ABSGRAD) {
          g[10] += fabsf(vx);
          g[11] += fabsf(vy);
        }
      }
      if (!contributed) continue;
      const float r = reduce_scatter12(g, b3, b2);
      if (slot >= 0) atomicAdd(v_splats + (int64_t)__float_as_int(q2.w) * REC + slot, r * post);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ tile schedule
// Longest-first launch order of the tiles.  A tile is one wave; its cost is its list length (forward estimate) or the
// number of entries the backward will actually walk (`work`, written by the forward).  Workgroups are dispatched in
// blockIdx order as slots free up, so feeding the heavy tiles first removes the tail where a few long tiles run alone
// on an otherwise idle chip.  Single workgroup, 1024-bucket counting sort; order inside a bucket is arbitrary (it
// affects speed only, never a result).
template <int ITEMS>  // register-cached costs per thread; n_tiles <= 1024 * ITEMS
__global__ __launch_bounds__(1024) void tile_order_kernel(int n_tiles, const int32_t* __restrict__ offsets,
                                                          int64_t n_isects, const int32_t* __restrict__ work,
                                                          int32_t* __restrict__ order, int32_t* __restrict__ n_long,
                                                          int32_t* n_long_post) {
  __shared__ int red[16];
  __shared__ unsigned hist[1024];
  __shared__ unsigned wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int cost[ITEMS];
  int m = 1;
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tid + i * 1024;
    cost[i] = 0;
    if (t < n_tiles) {
      if (work) {
        cost[i] = work[t];
      } else {
        const int e = (t + 1 < n_tiles) ? offsets[t + 1] : (int)n_isects;
        cost[i] = e - offsets[t];
      }
    }
    m = max(m, cost[i]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
  if (lane == 0) red[wave] = m;
  hist[tid] = 0;
  __syncthreads();
  m = red[0];
#pragma unroll
  for (int w = 1; w < 16; ++w) m = max(m, red[w]);
  // cost -> bucket, heavy first.  Buckets 0..255 hold exactly the costs above TILE_LONG (= 2048); below that the
  // bucket is 1023 - 3(c-1)/8, whose edges fall on c = 1024 | 1025 (bucket 640 | 639) and c = 512 | 513 (832 | 831),
  // so the size classes of gs_isect_tile_sort are contiguous ranges of the order
  static_assert(TILE_LONG == 2048, "bucket edges below assume 2048");
  const float scale_hi = 255.f / (float)max(m - TILE_LONG, 1);
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int c = max(cost[i], 0);
    cost[i] = (c > TILE_LONG) ? 255 - min(255, (int)((float)(c - TILE_LONG) * scale_hi))
                              : 1023 - ((3 * max(c - 1, 0)) >> 3);
    if (tid + i * 1024 < n_tiles) atomicAdd(&hist[cost[i]], 1u);
  }
  __syncthreads();
  const unsigned h = hist[tid];
  unsigned incl = h;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  unsigned before = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w)
    if (w < wave) before += wsum[w];
  hist[tid] = before + incl - h;
  if (n_long) {  // size_classes[3]: tiles with cost > 2048, > 1024, > 512 (= tiles in the buckets before 256, 640, 832)
    if (tid == 256) n_long[0] = (int32_t)(before + incl - h);
    if (tid == 640) n_long[1] = (int32_t)(before + incl - h);
    if (tid == 832) n_long[2] = (int32_t)(before + incl - h);
  }
  if (n_long_post) {  // the same three values straight into host memory the caller polls (gs_host_post_wait)
    const int slot = tid == 256 ? 0 : tid == 640 ? 1 : tid == 832 ? 2 : -1;
    if (slot >= 0)
      __hip_atomic_store(&n_long_post[slot], (int32_t)(before + incl - h), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < ITEMS; ++i) {
    const int t = tid + i * 1024;
    if (t < n_tiles) order[atomicAdd(&hist[cost[i]], 1u)] = t;
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

#define GS_RF_ARGS C, width, height, TW, TH, splats, isect_offsets, flatten_ids, n_isects, backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, tile_order, tile_work, sorted_counts, redo_flags, redo_pass, opacity_override, n_per_camera, final_T, n_dev
#define GS_RF_LAUNCH(CHN) hipLaunchKernelGGL((gs::rasterize_fwd_kernel<CHN>), grid, block, 0, s, GS_RF_ARGS)

static int rasterize_fwd_impl(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                              const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                              int64_t n_isects, const float* backgrounds, int32_t ed_last_channel,
                              float* render_colors, float* render_alphas, int32_t* last_ids,
                              const int32_t* tile_order, int32_t* tile_work, const int32_t* sorted_counts,
                              int32_t* redo_flags, int32_t redo_pass, const float* opacity_override,
                              int32_t n_per_camera, float* final_T, const int64_t* n_dev, void* stream) {
  int st = raster_check(C, width, height, tile_size, channels, n_isects);
  if (st != GS_OK) return st;
  if (!isect_offsets || !render_colors || !render_alphas || !last_ids) return GS_ERR_INVALID_ARG;
  if ((sorted_counts || redo_pass) && !redo_flags) return GS_ERR_INVALID_ARG;
  if (opacity_override && n_per_camera < 1) return GS_ERR_INVALID_ARG;
  if (n_isects > 0 && (!splats || !flatten_ids)) return GS_ERR_INVALID_ARG;
  if (gs_contract_take()) return GS_ERR_CONTRACT;  // a sort kernel found lists its tile_order did not describe
  const int TW = (width + 15) / 16, TH = (height + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(tile_order ? (unsigned)(C * TW * TH) : gs::raster_grid((unsigned)(C * TW * TH))), block(64);
  switch (channels) {
    case 1: GS_RF_LAUNCH(1); break;
    case 2: GS_RF_LAUNCH(2); break;
    case 3: GS_RF_LAUNCH(3); break;
    default: GS_RF_LAUNCH(4); break;
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_rasterize_fwd(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                                const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                int64_t n_isects, const float* backgrounds, int32_t ed_last_channel,
                                float* render_colors, float* render_alphas, int32_t* last_ids,
                                const int32_t* tile_order, int32_t* tile_work, const int32_t* sorted_counts,
                                int32_t* redo_flags, int32_t redo_pass, const float* opacity_override,
                                int32_t n_per_camera, float* final_T, void* stream) {
  return rasterize_fwd_impl(C, width, height, tile_size, channels, splats, isect_offsets, flatten_ids, n_isects,
                            backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, tile_order, tile_work,
                            sorted_counts, redo_flags, redo_pass, opacity_override, n_per_camera, final_T, nullptr,
                            stream);
}

// The same launch for a host that has not read the intersection count yet (include/gsplat_mi355x.h, "device-side
// n_isects"): `capacity` sizes flatten_ids, the count is read from *n_isects_dev by the kernel.
extern "C" int gs_rasterize_fwd_dev(int32_t C, int32_t width, int32_t height, int32_t tile_size, int32_t channels,
                                    const float* splats, const int32_t* isect_offsets, const int32_t* flatten_ids,
                                    int64_t capacity, const int64_t* n_isects_dev, const float* backgrounds,
                                    int32_t ed_last_channel, float* render_colors, float* render_alphas,
                                    int32_t* last_ids, const int32_t* tile_order, int32_t* tile_work,
                                    const float* opacity_override, int32_t n_per_camera, float* final_T, void* stream) {
  if (!n_isects_dev || capacity < 1 || !flatten_ids || !splats) return GS_ERR_INVALID_ARG;
  return rasterize_fwd_impl(C, width, height, tile_size, channels, splats, isect_offsets, flatten_ids, capacity,
                            backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, tile_order, tile_work,
                            nullptr, nullptr, 0, opacity_override, n_per_camera, final_T, n_isects_dev, stream);
}


#define GS_RB_ARGS C, width, height, TW, TH, splats, isect_offsets, flatten_ids, n_isects, backgrounds, ed_last_channel, render_colors, render_alphas, last_ids, v_render_colors, v_render_alphas, v_splats, tile_order, final_T
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
                                float* v_splats, const int32_t* tile_order, const float* final_T, void* stream) {
  int st = raster_check(C, width, height, tile_size, channels, n_isects);
  if (st != GS_OK) return st;
  if (n_isects == 0) return GS_OK;
  if (!splats || !isect_offsets || !flatten_ids || !render_colors || !render_alphas || !last_ids ||
      !v_render_colors || !v_render_alphas || !v_splats)
    return GS_ERR_INVALID_ARG;
  const int TW = (width + 15) / 16, TH = (height + 15) / 16;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(tile_order ? (unsigned)(C * TW * TH) : gs::raster_grid((unsigned)(C * TW * TH))), block(64);
  switch (channels) {
    case 1: GS_RB_LAUNCH(1); break;
    case 2: GS_RB_LAUNCH(2); break;
    case 3: GS_RB_LAUNCH(3); break;
    default: GS_RB_LAUNCH(4); break;
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_raster_tile_order_post(int32_t n_tiles, const int32_t* isect_offsets, int64_t n_isects,
                                         const int32_t* tile_work, int32_t* tile_order, int32_t* n_long,
                                         int32_t* n_long_post, void* stream) {
  if (n_tiles < 1 || n_isects < 0 || !tile_order) return GS_ERR_INVALID_ARG;
  if (!tile_work && !isect_offsets) return GS_ERR_INVALID_ARG;
  if (n_long_post && !n_long) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
#define GS_TO_ARGS n_tiles, isect_offsets, n_isects, tile_work, tile_order, n_long, n_long_post
  if (n_tiles <= 1024 * 10)
    hipLaunchKernelGGL(gs::tile_order_kernel<10>, dim3(1), dim3(1024), 0, s, GS_TO_ARGS);
  else if (n_tiles <= 1024 * 32)
    hipLaunchKernelGGL(gs::tile_order_kernel<32>, dim3(1), dim3(1024), 0, s, GS_TO_ARGS);
  else if (n_tiles <= 1024 * 128)
    hipLaunchKernelGGL(gs::tile_order_kernel<128>, dim3(1), dim3(1024), 0, s, GS_TO_ARGS);
  else
    return GS_ERR_UNSUPPORTED;
#undef GS_TO_ARGS
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_raster_tile_order(int32_t n_tiles, const int32_t* isect_offsets, int64_t n_isects,
                                    const int32_t* tile_work, int32_t* tile_order, int32_t* n_long, void* stream) {
  return gs_raster_tile_order_post(n_tiles, isect_offsets, n_isects, tile_work, tile_order, n_long, nullptr, stream);
}
