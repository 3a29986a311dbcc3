// isect_tiles + sort + isect_offset_encode as ONE binned pipeline for gfx950.
//
// gsplat emits one 64-bit (camera|tile|depth) key per intersection and radix-sorts all of them (6 sweeps of a
// 12-B/element array at config 2).  The result is fully determined by three facts: which (row, tile) pairs exist,
// the row depths, and the tie rule (equal keys keep ascending flat row id).  So build it directly:
//   1. gs_isect_bin_count    every workgroup owns one camera and a slice of its rows and counts tile hits in an LDS
//                            histogram over that camera's tiles  ->  table[camera][slice][tile]
//   2. gs_isect_bin_offsets  per tile: exclusive prefix over the slices, total per tile; exclusive scan of the totals
//                            = isect_offsets (A.4's lower bounds, no search needed), last value = n_isects
//   3. gs_isect_bin_scatter_regions   a workgroup owns a band of tile rows and a group of slices, walks the visible
//                            lists the count pass left (16 B per visible row) clipped to its band; an LDS cursor per
//                            tile (offset + slice prefix) hands every hit its slot in the tile's segment; writes
//                            (depth bits, flat row id) pairs, order inside a segment arbitrary.  (gs_isect_bin_scatter:
//                            the same by row slice, from the records, for callers without the lists.)
//   4. gs_isect_tile_sort    one workgroup per tile segment, sorted IN LDS on the 64-bit (depth, row id) key: counting
//                            partition on a linear-depth bucket + per-bucket insertion sort, LSD radix fallback for
//                            crowded buckets; writes the final isect_ids / flatten_ids.  Segments beyond the LDS
//                            capacity run the same code on global scratch.  gs_isect_tile_sort_front sorts only the
//                            leading entries of long segments (see "front sort" below).
// Bit-exact with A.3/A.4 (same keys, same order, same offsets); the atomics only decide the order INSIDE a tile
// segment before step 4, which the sort then fixes completely.  HBM traffic: 2 reads of the visible records,
// one 8-B write + read and one 12-B write per intersection.
#include "gs_common.hpp"
#include "raster_cull.hpp"

namespace gs {

constexpr int BIN_THREADS = 1024;      // 16 waves: a quarter as many slices as with 256 threads, so the scatter's
                                       // (slice, tile) runs are 4x longer (0.129 -> 0.106 ms at config 2)
constexpr int BIN_COOP = 16;            // tile boxes larger than this are walked by the whole wave
constexpr int BIN_MAX_TILES = 16384;    // per camera: LDS histogram of 64 KiB

// ------------------------------------------------------------------------------------------ count / scatter
// A workgroup streams its row slice 1024 rows at a time.  Most rows of a camera are culled (tiles_per_gauss == 0),
// so the visible ones are first compacted into an LDS queue {row id, tile box, depth bits} and the tile walk then
// runs on full waves.  cnt[] is the camera's tile histogram (count) or the per-tile write cursor (scatter).

// CULL (count pass only): a (row, tile) pair is counted only if the splat can reach alpha >= 1/255 on some pixel of
// the tile (rect_may_touch over the tile's 16x16 pixel centres, the rasterizer's own whole-tile test).  Dropping the
// others changes no output bit of the rasterizer -- every pixel skips them (A.5) -- but shortens the lists the
// scatter writes, the sort orders and both rasterizer passes stage.  The decision is taken ONCE, here: a box of up to
// 64 tiles leaves a bit mask (bit t = tile t of the box in row-major order) beside its visible-list entry and the
// scatter follows the mask, so the two passes agree by construction; larger boxes (rare) are not culled.
// t / w and t % w for 0 <= t < 2^14, 1 <= w <= 2^14 through a float reciprocal (`rw` ~ 1 / w to an ulp): (t + 1/2) / w is
// at least 1 / (2 w) away from every integer, far more than the product's rounding error, so the truncation is exact.
// (The integer division the compiler emits instead is ~25 instructions per quotient.)
__device__ __forceinline__ void div_small(int t, int w, float rw, int& q, int& r) {
  q = (int)(((float)t + 0.5f) * rw);
  r = t - q * w;
}

// value of lane `src` (wave-uniform): a v_readlane into a scalar register where __shfl would go through the LDS crossbar
__device__ __forceinline__ int lane_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }
__device__ __forceinline__ uint32_t lane_u(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ float lane_f(float v, int src) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}

struct CullRec {
  float mx, my, A, Bh, C, op;
};

// One wave walks the tile boxes of up to 64 visible-list entries, one per lane (`valid`): histogram / cursor updates in
// `cnt`.  Returns the lane's tile mask (CULL).
template <bool SCATTER, bool CULL>
__device__ __forceinline__ unsigned long long bin_walk(uint32_t* cnt, bool valid, const uint4 en, const CullRec g,
                                                       int tile_w, float tile_px, uint2* __restrict__ pairs) {
  const int lane = threadIdx.x & 63;
  {
    int cntg = 0, x0 = 0, y0 = 0, x1 = 0;
    uint32_t idx = 0, dbits = 0;
    if (valid) {
      idx = en.x; dbits = en.w;
      x0 = en.y & 0xffff; y0 = en.y >> 16; x1 = en.z & 0xffff;
      cntg = (int)((en.z >> 16) - y0) * (x1 - x0);
    }
    unsigned long long mask = 0ull;
    if (!CULL && cntg > 0 && cntg <= BIN_COOP) {
      int tj = x0, trow0 = y0 * tile_w;
      for (int t = 0; t < cntg; ++t) {
        const uint32_t pos = atomicAdd(&cnt[trow0 + tj], 1u);
        if (SCATTER) pairs[pos] = make_uint2(dbits, idx);
        if (++tj == x1) { tj = x0; trow0 += tile_w; }
      }
    }
    if (CULL && cntg > 0 && cntg <= BIN_COOP) {
      // Small box, one lane: per tile ROW the columns the splat can reach.  The region {alpha >= 1/255} is the ellipse
      // pw(dx, dy) >= L, L = log2(tau / opacity) (minus the slack of rect_may_touch, taken over the whole box); cut by
      // the row's strip of pixel centres it is convex, so its x-extent is an interval whose ends are reached at the
      // ellipse's x-extreme points (dy = -+ Bh xmax / C) clamped into the strip.  Same decision as rect_may_touch per
      // tile up to rounding, both conservative (0.01 px margin here); a row costs one evaluation instead of one per tile.
      const float ext = tile_px - 1.f;
      const int w = x1 - x0, y1 = y0 + cntg / w;
      const float Dx = fmaxf(fabsf(g.mx - ((float)x0 * tile_px + 0.5f)), fabsf(g.mx - ((float)x1 * tile_px - 0.5f)));
      const float Dy = fmaxf(fabsf(g.my - ((float)y0 * tile_px + 0.5f)), fabsf(g.my - ((float)y1 * tile_px - 0.5f)));
      const float slack = 1e-5f * (fabsf(g.A) * Dx * Dx + fabsf(g.C) * Dy * Dy) + 1e-2f;
      const float L = __builtin_amdgcn_logf(ALPHA_MIN * 0.999f * __builtin_amdgcn_rcpf(g.op)) - slack;  // log2
      const float det = g.A * g.C - g.Bh * g.Bh;
      // a record the projection never writes for a visible row (NaN, non-definite conic) -> every tile, as rect_may_touch
      const bool sane = (g.A < 0.f) && (g.C < 0.f) && (det > 0.f) && (g.op > 0.f) && (L == L) && (fabsf(L) < 3.0e38f);
      const float rdet = __builtin_amdgcn_rcpf(det), rA = __builtin_amdgcn_rcpf(g.A);
      const float ymax2 = g.A * L * rdet;
      const float xmax = __builtin_amdgcn_sqrtf(fmaxf(0.f, g.C * L * rdet));
      const float dyr = -g.Bh * xmax * __builtin_amdgcn_rcpf(g.C);
      const float rtile = __builtin_amdgcn_rcpf(tile_px);
      // an opacity that cannot reach 1/255 even at the mean (zeroed instances, rigid.py:493): no tile takes the pair,
      // whatever the conic looks like (rect_may_touch says the same: opacity * exp2(<= 0) stays under the threshold)
      const bool never = g.op < ALPHA_MIN * 0.999f;
      int trow0 = y0 * tile_w, sh = 0;
      for (int ti = y0; ti < (never ? y0 : y1); ++ti, trow0 += tile_w, sh += w) {
        int lo = x0, hi = x1 - 1;
        if (sane) {
          const float dyh = g.my - ((float)ti * tile_px + 0.5f), dyl = dyh - ext;
          const float dyn = fminf(fmaxf(0.f, dyl), dyh);
          const float dy1 = fminf(fmaxf(dyr, dyl), dyh), dy2 = fminf(fmaxf(-dyr, dyl), dyh);
          const float r = -g.Bh * dy1 * rA - __builtin_amdgcn_sqrtf(fmaxf(0.f, g.A * L - dy1 * dy1 * det)) * rA;
          const float l = -g.Bh * dy2 * rA + __builtin_amdgcn_sqrtf(fmaxf(0.f, g.A * L - dy2 * dy2 * det)) * rA;
          const float m = 1e-2f + 1e-6f * (fabsf(r) + fabsf(l) + fabsf(g.mx));
          const float xlo = g.mx - r - m, xhi = g.mx - l + m;  // pixel centres the splat can reach in this row
          lo = max(x0, (int)ceilf((xlo - 0.5f - ext) * rtile));
          hi = min(x1 - 1, (int)floorf((xhi - 0.5f) * rtile));
          if (!(L < 0.f) || dyn * dyn > ymax2 * 1.0001f + 1e-3f) hi = lo - 1;
        }
        for (int tj = lo; tj <= hi; ++tj) {
          atomicAdd(&cnt[trow0 + tj], 1u);
          mask |= 1ull << (sh + tj - x0);
        }
      }
    }
    unsigned long long big = __ballot(cntg > BIN_COOP);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const int b_cnt = lane_i(cntg, src), b_x0 = lane_i(x0, src), b_y0 = lane_i(y0, src);
      const int b_w = lane_i(x1, src) - b_x0;
      const float b_rw = __builtin_amdgcn_rcpf((float)b_w);
      const uint32_t b_d = lane_u(dbits, src), b_idx = lane_u(idx, src);
      if (CULL && b_cnt <= 64) {
        const float mx = lane_f(g.mx, src), my = lane_f(g.my, src), A = lane_f(g.A, src), Bh = lane_f(g.Bh, src);
        const float Cc = lane_f(g.C, src), op = lane_f(g.op, src);
        int dq, dr;
        div_small(lane, b_w, b_rw, dq, dr);
        const int ti = b_y0 + dq, tj = b_x0 + dr;
        const bool hit = lane < b_cnt && rect_may_touch(mx, my, A, Bh, Cc, op, (float)tj * tile_px + 0.5f,
                                                        (float)ti * tile_px + 0.5f, tile_px - 1.f);
        const unsigned long long bal = __ballot(hit);
        if (hit) atomicAdd(&cnt[ti * tile_w + tj], 1u);
        if (lane == src) mask = bal;
        continue;
      }
      for (int t = lane; t < b_cnt; t += 64) {
        int dq, dr;
        div_small(t, b_w, b_rw, dq, dr);
        const uint32_t pos = atomicAdd(&cnt[(b_y0 + dq) * tile_w + b_x0 + dr], 1u);
        if (SCATTER) pairs[pos] = make_uint2(b_d, b_idx);
      }
    }
    return mask;
  }
}

// Every WAVE keeps a private ring of 128 queue slots and walks the tile boxes of 64 entries whenever it holds that many:
// the walk always runs on full waves however few rows of a step are visible, and no workgroup barrier is needed between
// the first and the last step (a workgroup-wide queue drained every 2048 rows filled 6 of 16 waves at 20 % visibility
// and cost three barriers per drain).  The visible-list segment of the slice is appended 64 entries at a time, in
// whatever order the waves get there; the order of a slice's list only decides the slot of a pair inside its tile
// segment before the sort, which fixes it.
constexpr int BIN_RING = 128;

template <bool SCATTER, bool CULL>
__global__ __launch_bounds__(BIN_THREADS) void isect_bin_kernel(int32_t N, int32_t tile_w, int32_t T, int32_t n_slices,
                                                                int32_t rows_per_slice,
                                                                const float* __restrict__ splats,
                                                                const int32_t* __restrict__ tiles_per_gauss,
                                                                uint32_t* __restrict__ table,
                                                                const int32_t* __restrict__ offsets,
                                                                uint2* __restrict__ pairs,
                                                                uint4* __restrict__ vis_list,
                                                                int32_t* __restrict__ vis_cnt,
                                                                uint2* __restrict__ vis_mask, float tile_px) {
  static_assert(!(SCATTER && CULL), "the cull decision is taken by the count pass");
  extern __shared__ uint32_t cnt[];  // [T]
  __shared__ uint4 q[BIN_THREADS / 64][BIN_RING];
  __shared__ int s_written;
  const int c = blockIdx.y, slice = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t* trow = table + ((int64_t)c * n_slices + slice) * T;
  for (int t = tid; t < T; t += BIN_THREADS) cnt[t] = SCATTER ? (uint32_t)offsets[(int64_t)c * T + t] + trow[t] : 0u;
  if (tid == 0) s_written = 0;
  __syncthreads();
  const int n0 = slice * rows_per_slice;
  const int n1 = min(N, n0 + rows_per_slice);
  // count pass only: the slice's queue entries also go to its segment of vis_list, which the region scatter reads
  const bool keep = !SCATTER && vis_list != nullptr;
  uint4* seg = keep ? vis_list + ((int64_t)c * n_slices + slice) * rows_per_slice : nullptr;
  uint2* mseg = CULL ? vis_mask + ((int64_t)c * n_slices + slice) * rows_per_slice : nullptr;
  uint4* qw = q[tid >> 6];
  int head = 0, tail = 0;  // ring positions (monotone; slot = position & (BIN_RING - 1)), wave-uniform

  auto drain = [&](int n) {  // walk ring entries [head, head + n), n <= 64
    const bool valid = lane < n;
    uint4 en = make_uint4(0u, 0u, 0u, 0u);
    CullRec g = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (valid) {
      en = qw[(head + lane) & (BIN_RING - 1)];
      if (CULL) {  // the record's mean / conic / opacity: its line was read for the box a few steps ago
        const float4* rp = reinterpret_cast<const float4*>(splats + (int64_t)en.x * REC);
        const float4 r0 = rp[0], r1 = rp[1];
        g = {r0.x, r0.y, CONIC_SCALE * r0.z, CONIC_SCALE * r0.w, CONIC_SCALE * r1.x, r1.y};
      }
    }
    int base = 0;
    if (keep) {
      if (lane == 0) base = atomicAdd(&s_written, n);
      base = __shfl(base, 0);
      if (valid) seg[base + lane] = en;
    }
    const unsigned long long mask = bin_walk<SCATTER, CULL>(cnt, valid, en, g, tile_w, tile_px, pairs);
    if (CULL && valid) mseg[base + lane] = make_uint2((uint32_t)mask, (uint32_t)(mask >> 32));
    head += n;
  };

  // software pipeline over the 1024-row steps: a step's tile counts are loaded two steps ahead and its visible rows'
  // boxes one step ahead, so the gathers of step k+1 are in flight while step k is queued and walked
  auto count_of = [&](int row) { return row < n1 ? tiles_per_gauss[(int64_t)c * N + row] : 0; };
  int cntg = count_of(n0 + tid), cnt_next = count_of(n0 + BIN_THREADS + tid);
  float4 r3 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t dbits = 0u;
  if (cntg > 0) {
    const int64_t i0 = (int64_t)c * N + n0 + tid;
    r3 = reinterpret_cast<const float4*>(splats + i0 * REC)[3];
    if (SCATTER || keep) dbits = (uint32_t)__float_as_int(splats[i0 * REC + 10]);
  }
  for (int nb = n0; nb < n1; nb += BIN_THREADS) {
    const int64_t idx = (int64_t)c * N + nb + tid;
    const bool vis = cntg > 0;
    float4 r3_next = make_float4(0.f, 0.f, 0.f, 0.f);
    uint32_t dbits_next = 0u;
    if (cnt_next > 0) {
      const int64_t i1 = idx + BIN_THREADS;
      r3_next = reinterpret_cast<const float4*>(splats + i1 * REC)[3];
      if (SCATTER || keep) dbits_next = (uint32_t)__float_as_int(splats[i1 * REC + 10]);
    }
    const int cnt_nn = count_of(nb + 2 * BIN_THREADS + tid);
    const unsigned long long vm = __ballot(vis);
    if (vis)
      qw[(tail + __popcll(vm & lt_mask)) & (BIN_RING - 1)] =
          make_uint4((uint32_t)idx, (uint32_t)__float_as_int(r3.y), (uint32_t)__float_as_int(r3.z), dbits);
    tail += __popcll(vm);
    cntg = cnt_next; cnt_next = cnt_nn; r3 = r3_next; dbits = dbits_next;
    // (the ring is written and read by this wave only: LDS operations of one wave complete in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tail - head >= 64) drain(64);  // at most 63 + 64 entries are ever pending: one walk per step suffices
  }
  if (tail > head) drain(tail - head);
  if (!SCATTER) {
    __syncthreads();
    for (int t = tid; t < T; t += BIN_THREADS) trow[t] = cnt[t];
    if (keep && tid == 0) vis_cnt[c * n_slices + slice] = s_written;
  }
}

// Scatter by tile REGION.  isect_bin_kernel<true> gives a workgroup one row slice and all of the camera's tiles, so the
// run it writes into one tile segment is (hits of the slice) / T entries long -- 8 at config 2, 2 at 9600 tiles: 8-B
// stores spread over as many partially written lines as there are tiles, most of which leave L2 before they fill.
// Here a workgroup owns a band of tile rows (1 / R of the tiles) and a GROUP of G consecutive slices, walks the group's
// visible lists (written by the count pass: 16 B per visible row, contiguous) clipped to its band, and so writes runs
// G * R times as long from a cursor table R times smaller.  Which slot a hit gets inside its tile segment differs from
// the slice scatter's; the per-tile sort that follows fixes the order completely either way.
// CULL: the count pass's tile masks (see bin_drain) say which tiles of a box of <= 64 tiles take the pair.
// The walk of the visible lists of slices [sa, sb) (sa < sb) of one camera, clipped to the band [ty0, ty1):
// `emit(band tile, depth bits, row)` per pair.  A wave takes every 16th batch of 64 entries of each slice in turn and
// requests its NEXT batch (entry + mask, across slice ends) before it walks the current one, so the two loads at the
// head of a batch are not waited for in sequence (worth 1-2 us of 48 at 1242x375 x 5).
template <bool CULL, class Emit>
__device__ __forceinline__ void region_walk(int sa, int sb, const int32_t* __restrict__ cnt_row,
                                            const uint4* __restrict__ list_row, const uint2* __restrict__ mask_row,
                                            int seg_cap, int ty0, int ty1, int tile_w, Emit emit) {
  const int tid = threadIdx.x, lane = tid & 63, wave0 = (tid >> 6) * 64;
  int s = sa, e0 = wave0;  // the batch about to be fetched (wave-uniform)
  int n = cnt_row[s], n_next = s + 1 < sb ? cnt_row[s + 1] : 0;
  auto settle = [&]() {  // skip to the next slice that still has a batch for this wave
    while (s < sb && e0 >= n) {
      ++s;
      e0 = wave0;
      n = n_next;
      n_next = s + 1 < sb ? cnt_row[s + 1] : 0;
    }
  };
  uint4 en_n = make_uint4(0u, 0u, 0u, 0u);
  uint2 mm_n = make_uint2(0u, 0u);
  bool v_n = false;
  auto fetch = [&]() {
    v_n = s < sb && e0 + lane < n;
    if (v_n) {
      const int64_t at = (int64_t)s * seg_cap + e0 + lane;
      en_n = list_row[at];
      if (CULL) mm_n = mask_row[at];
    }
  };
  settle();
  bool have = s < sb;
  fetch();
  while (have) {
    const uint4 en = en_n;
    const uint2 mm = mm_n;
    const bool valid = v_n;
    e0 += BIN_THREADS;
    settle();
    have = s < sb;
    fetch();
    int cntg = 0, x0 = 0, y0 = 0, x1 = 0;
    uint32_t idx = 0, dbits = 0;
    unsigned long long m = 0ull;  // CULL: the band's part of the tile mask, bit 0 = tile (x0, y0)
    bool masked = false;
    if (valid) {
      idx = en.x; dbits = en.w;
      x0 = en.y & 0xffff; x1 = en.z & 0xffff;
      const int y0o = (int)(en.y >> 16), y1o = (int)(en.z >> 16), w = x1 - x0;
      y0 = max(y0o, ty0);
      const int rows = max(0, min(y1o, ty1) - y0);
      cntg = rows * w;
      if (CULL && (y1o - y0o) * w <= 64) {
        masked = true;
        if (cntg > 0) {
          m = (((unsigned long long)mm.y << 32) | mm.x) >> ((y0 - y0o) * w);
          if (cntg < 64) m &= (1ull << cntg) - 1ull;
        }
        cntg = __popcll(m);  // pairs this lane will write
      }
    }
    if (cntg > 0 && cntg <= BIN_COOP) {
      if (CULL && masked) {
        // one flat loop over the set bits (a loop over the rows with one over the row's bits inside runs, wave-wide,
        // the sum over rows of the longest row): bit j -> (row, column) of the box by a float reciprocal, exact for
        // j < 64, w <= 64 ((j + 1/2) / w stays 1/128 away from every integer)
        const int w = x1 - x0;
        const float rw = __builtin_amdgcn_rcpf((float)w);
        const int base = (y0 - ty0) * tile_w + x0;
        while (m) {
          const int j = __ffsll((long long)m) - 1;
          m &= m - 1;
          int q, rem;
          div_small(j, w, rw, q, rem);
          emit(base + q * tile_w + rem, dbits, idx);
        }
      } else {
        int tj = x0, trow0 = (y0 - ty0) * tile_w;
        for (int t = 0; t < cntg; ++t) {
          emit(trow0 + tj, dbits, idx);
          if (++tj == x1) { tj = x0; trow0 += tile_w; }
        }
      }
    }
    unsigned long long big = __ballot(cntg > BIN_COOP);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const int b_cnt = lane_i(cntg, src), b_x0 = lane_i(x0, src), b_y0 = lane_i(y0, src) - ty0;
      const int b_w = lane_i(x1, src) - b_x0;
      const float b_rw = __builtin_amdgcn_rcpf((float)b_w);
      const uint32_t b_d = lane_u(dbits, src), b_idx = lane_u(idx, src);
      int dq, dr;
      if (CULL && lane_i((int)masked, src)) {
        const uint32_t mlo = lane_u((uint32_t)m, src), mhi = lane_u((uint32_t)(m >> 32), src);
        const unsigned long long bm = ((unsigned long long)mhi << 32) | mlo;
        div_small(lane, b_w, b_rw, dq, dr);
        if ((bm >> lane) & 1ull) emit((b_y0 + dq) * tile_w + b_x0 + dr, b_d, b_idx);
        continue;
      }
      for (int t = lane; t < b_cnt; t += 64) {
        div_small(t, b_w, b_rw, dq, dr);
        emit((b_y0 + dq) * tile_w + b_x0 + dr, b_d, b_idx);
      }
    }
  }
}

template <bool CULL>
__global__ __launch_bounds__(BIN_THREADS) void isect_region_scatter_kernel(
    int32_t tile_w, int32_t tile_h, int32_t T, int32_t n_slices, int32_t G, int32_t R, int32_t seg_cap,
    const uint4* __restrict__ vis_list, const int32_t* __restrict__ vis_cnt, const uint2* __restrict__ vis_mask,
    const uint32_t* __restrict__ table, const int32_t* __restrict__ offsets, uint2* __restrict__ pairs,
    const int64_t* __restrict__ n_dev, int64_t pair_cap) {
  extern __shared__ uint32_t cur[];  // [band tiles]
  if (n_dev && *n_dev > pair_cap) return;  // speculative capacity too small: the host redoes the call (gs_isect_bin_build_dev)
  const int c = blockIdx.y, g = blockIdx.x / R, r = blockIdx.x - g * R;
  const int tid = threadIdx.x;
  const int ty0 = (int)((int64_t)r * tile_h / R), ty1 = (int)((int64_t)(r + 1) * tile_h / R);
  const int t0 = ty0 * tile_w, nt = (ty1 - ty0) * tile_w;
  if (nt == 0) return;
  const int s0 = g * G, s1 = min(n_slices, s0 + G);
  // table holds the exclusive prefix over slices: the group's first slice marks where the group's hits start
  const uint32_t* trow = table + ((int64_t)c * n_slices + s0) * T + t0;
  const int32_t* orow = offsets + (int64_t)c * T + t0;
  for (int t = tid; t < nt; t += BIN_THREADS) cur[t] = (uint32_t)orow[t] + trow[t];
  __syncthreads();
  region_walk<CULL>(s0, s1, vis_cnt + (int64_t)c * n_slices, vis_list + (int64_t)c * n_slices * seg_cap,
                    CULL ? vis_mask + (int64_t)c * n_slices * seg_cap : nullptr, seg_cap, ty0, ty1, tile_w,
                    [&](int t, uint32_t dbits, uint32_t idx) {
                      const uint32_t pos = atomicAdd(&cur[t], 1u);
                      pairs[pos] = make_uint2(dbits, idx);
                    });
}

// The same scatter STAGED through LDS.  The direct form above issues one 8-B store per pair from lanes that sit in
// different tiles: ~10 distinct lines per store instruction, each line dribbled out over the whole walk and most of them
// leaving L2 partially written (3.2x write amplification, profiles/traffic.json r02_v3).  Here the workgroup knows how
// many pairs its slices put into every tile of its band (a difference of two rows of the slice-prefix table), so it
// counting-sorts them by tile in LDS first and then writes each tile's run in one piece, consecutive lanes on
// consecutive entries.  The slices of the group are taken in chunks whose pairs fit the staging area (halving the chunk
// until they do; a single slice that still does not fit is scattered directly).
// What it buys (profiles/r02_ab_runs.txt item 12): long runs make ONE band per camera affordable at 1242x375, which
// halves the list reads and the lanes that idle on out-of-band entries (48.1 -> 42.8 us); at 9 600 tiles the direct
// form's partial lines no longer thrash L2 (113 -> 94 us).  With the same bands as the direct form it is no faster:
// with emission switched off altogether the walk alone takes 34 of the 48 us (10.7 wave-iterations per 64 entries at
// 23 % lane use: the longest mask of the batch sets the trip count) -- the stores were never the larger part.
template <bool CULL>
__global__ __launch_bounds__(BIN_THREADS) void isect_region_scatter_staged_kernel(
    int32_t tile_w, int32_t tile_h, int32_t T, int32_t n_slices, int32_t G, int32_t R, int32_t seg_cap,
    int32_t nt_max, int32_t cap, const uint4* __restrict__ vis_list, const int32_t* __restrict__ vis_cnt,
    const uint2* __restrict__ vis_mask, const uint32_t* __restrict__ table, const int32_t* __restrict__ offsets,
    const int32_t* __restrict__ tile_counts, uint2* __restrict__ pairs, const int64_t* __restrict__ n_dev,
    int64_t pair_cap) {
  extern __shared__ uint32_t lds_raw[];
  if (n_dev && *n_dev > pair_cap) return;  // (see isect_region_scatter_kernel)
  uint32_t* lcur = lds_raw;                                        // [nt_max] staging cursor per band tile
  uint32_t* gdelta = lds_raw + nt_max;                             // [nt_max] global position - staging position
  uint2* stage = reinterpret_cast<uint2*>(lds_raw + 2 * nt_max);  // [cap] {depth bits, row}
  uint16_t* tl = reinterpret_cast<uint16_t*>(stage + cap);        // [cap] band tile of the staged entry
  __shared__ uint32_t wsum[BIN_THREADS / 64];
  const int c = blockIdx.y, g = blockIdx.x / R, r = blockIdx.x - g * R;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ty0 = (int)((int64_t)r * tile_h / R), ty1 = (int)((int64_t)(r + 1) * tile_h / R);
  const int t0 = ty0 * tile_w, nt = (ty1 - ty0) * tile_w;
  if (nt == 0) return;
  const int s0 = g * G, s1 = min(n_slices, s0 + G);
  const int32_t* orow = offsets + (int64_t)c * T + t0;
  const int32_t* crow = tile_counts + (int64_t)c * T + t0;
  const int K = (nt + BIN_THREADS - 1) / BIN_THREADS;  // consecutive tiles per thread in the scan
  int sa = s0;
  while (sa < s1) {
    const uint32_t* ra = table + ((int64_t)c * n_slices + sa) * T + t0;
    int sb = s1;
    uint32_t total;
    for (;;) {
      // pairs per tile of slices [sa, sb): difference of the prefix rows (past the last slice: the tile's total)
      const uint32_t* rb = sb < n_slices ? table + ((int64_t)c * n_slices + sb) * T + t0 : nullptr;
      for (int t = tid; t < nt; t += BIN_THREADS) lcur[t] = (rb ? rb[t] : (uint32_t)crow[t]) - ra[t];
      __syncthreads();
      uint32_t mine = 0;
      const int k0 = tid * K, k1 = min(nt, k0 + K);
      for (int t = k0; t < k1; ++t) mine += lcur[t];
      uint32_t incl = mine;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      uint32_t before = 0, all = 0;
#pragma unroll
      for (int w = 0; w < BIN_THREADS / 64; ++w) {
        if (w < wave) before += wsum[w];
        all += wsum[w];
      }
      total = all;
      if (total <= (uint32_t)cap || sb - sa == 1) {
        uint32_t run = before + incl - mine;
        for (int t = k0; t < k1; ++t) {
          const uint32_t v = lcur[t];
          lcur[t] = run;
          run += v;
        }
        break;
      }
      sb = sa + (sb - sa) / 2;
      __syncthreads();  // (wsum and lcur are rewritten by the next attempt)
    }
    __syncthreads();
    const bool direct = total > (uint32_t)cap;  // one slice with more pairs than the staging area holds
    for (int t = tid; t < nt; t += BIN_THREADS) {
      const uint32_t gpos = (uint32_t)orow[t] + ra[t];
      if (direct) lcur[t] = gpos; else gdelta[t] = gpos - lcur[t];
    }
    __syncthreads();
    const int32_t* cnt_row = vis_cnt + (int64_t)c * n_slices;
    const uint4* list_row = vis_list + (int64_t)c * n_slices * seg_cap;
    const uint2* mask_row = CULL ? vis_mask + (int64_t)c * n_slices * seg_cap : nullptr;
    if (direct)
      region_walk<CULL>(sa, sb, cnt_row, list_row, mask_row, seg_cap, ty0, ty1, tile_w,
                        [&](int t, uint32_t dbits, uint32_t idx) {
                          const uint32_t pos = atomicAdd(&lcur[t], 1u);
                          pairs[pos] = make_uint2(dbits, idx);
                        });
    else
      region_walk<CULL>(sa, sb, cnt_row, list_row, mask_row, seg_cap, ty0, ty1, tile_w,
                        [&](int t, uint32_t dbits, uint32_t idx) {
                          const uint32_t pos = atomicAdd(&lcur[t], 1u);
                          stage[pos] = make_uint2(dbits, idx);
                          tl[pos] = (uint16_t)t;
                        });
    __syncthreads();
    if (!direct)
      for (uint32_t e = tid; e < total; e += BIN_THREADS) pairs[gdelta[tl[e]] + e] = stage[e];
    __syncthreads();
    sa = sb;
  }
}

// per tile: exclusive prefix over the slices (in place), total -> counts.  A workgroup owns 64 consecutive tiles; its
// four waves split the slices (sum their quarter, exchange the quarter totals through LDS, then write the prefixes), so
// the serial chain per tile is n_slices / 4 long and there are 4x the waves to hide it (one wave per 64 tiles left
// most CUs empty: 0.056 -> see DESIGN.md at 9600 tiles x 512 slices).
__global__ __launch_bounds__(256) void bin_slice_prefix_kernel(int32_t C, int32_t T, int32_t n_slices,
                                                               uint32_t* __restrict__ table,
                                                               int32_t* __restrict__ counts) {
  __shared__ uint32_t part[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t g = (int64_t)blockIdx.x * 64 + lane;
  const bool live = g < (int64_t)C * T;
  const int c = live ? (int)(g / T) : 0, t = live ? (int)(g - (int64_t)c * T) : 0;
  uint32_t* col = table + (int64_t)c * n_slices * T + t;
  const int per = (n_slices + 3) / 4;
  const int s0 = min(n_slices, w * per), s1 = min(n_slices, s0 + per);
  uint32_t sum = 0;
  if (live) {
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      const uint32_t v0 = col[(int64_t)(s + 0) * T], v1 = col[(int64_t)(s + 1) * T];
      const uint32_t v2 = col[(int64_t)(s + 2) * T], v3 = col[(int64_t)(s + 3) * T];
      sum += (v0 + v1) + (v2 + v3);
    }
    for (; s < s1; ++s) sum += col[(int64_t)s * T];
  }
  part[w][lane] = sum;
  __syncthreads();
  if (!live) return;
  uint32_t run = 0;
  for (int k = 0; k < w; ++k) run += part[k][lane];
  if (w == 3) counts[g] = (int32_t)(run + sum);
  int s = s0;
  for (; s + 4 <= s1; s += 4) {
    const uint32_t v0 = col[(int64_t)(s + 0) * T], v1 = col[(int64_t)(s + 1) * T];
    const uint32_t v2 = col[(int64_t)(s + 2) * T], v3 = col[(int64_t)(s + 3) * T];
    col[(int64_t)(s + 0) * T] = run; run += v0;
    col[(int64_t)(s + 1) * T] = run; run += v1;
    col[(int64_t)(s + 2) * T] = run; run += v2;
    col[(int64_t)(s + 3) * T] = run; run += v3;
  }
  for (; s < s1; ++s) {
    const uint32_t v = col[(int64_t)s * T];
    col[(int64_t)s * T] = run;
    run += v;
  }
}

// single workgroup: exclusive scan of the per-tile totals -> offsets, grand total -> *total.  ITEMS consecutive tiles per
// thread and round; the host picks the smallest ITEMS that covers n_tiles in ONE round (each round is a chain of three
// barriers on the path to the host's n_isects read)
template <int ITEMS>
__global__ __launch_bounds__(1024) void bin_offsets_kernel(int64_t n_tiles, const int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ offsets, int64_t* __restrict__ total,
                                                           int64_t* total_post, int32_t* classes_post) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  __shared__ int32_t cls_s[3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  if (tid < 3) cls_s[tid] = 0;
  __syncthreads();
  int32_t n_gt[3] = {0, 0, 0};  // this thread's lists longer than 2048 / 1024 / 512 entries (the sort's size classes)
  for (int64_t base = 0; base < n_tiles; base += 1024 * ITEMS) {
    const int64_t i0 = base + (int64_t)tid * ITEMS;
    int32_t v[ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      v[i] = (i0 + i < n_tiles) ? counts[i0 + i] : 0;
      s += v[i];
      n_gt[0] += v[i] > TILE_LONG;
      n_gt[1] += v[i] > 1024;
      n_gt[2] += v[i] > 512;
    }
    int64_t incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int64_t before = carry_s, all = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) before += wsum[w];
      all += wsum[w];
    }
    int64_t ex = before + incl - s;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      if (i0 + i < n_tiles) offsets[i0 + i] = (int32_t)ex;
      ex += v[i];
    }
    __syncthreads();
    if (tid == 0) carry_s += all;
    __syncthreads();
  }
  if (classes_post) {
    // the size classes gs_raster_tile_order will find (#lists > 2048, > 1024, > 512), posted from here: the host has
    // them with n_isects, one kernel earlier than the tile order could give them
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int32_t n = n_gt[c];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
      if (lane == 0 && n) atomicAdd(&cls_s[c], n);
    }
    __syncthreads();
    if (tid < 3) __hip_atomic_store(&classes_post[tid], cls_s[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (tid == 0) {
    *total = carry_s;
    // ... and straight into host memory the caller polls (gs_host_post_wait): no copy, no stream synchronisation
    if (total_post) __hip_atomic_store(total_post, carry_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ------------------------------------------------------------------------------------------ per-tile sort
constexpr int TS_MAX_BITS = 9;
constexpr int TS_MAX_RADIX = 1 << TS_MAX_BITS;

// lanes (among the valid ones) holding the same digit (up to 9 bits)
__device__ __forceinline__ unsigned long long ts_peers(uint32_t d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < TS_MAX_BITS; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

template <int WAVES>
struct TsShared {
  uint32_t wave_cnt[WAVES][TS_MAX_RADIX];
  uint32_t wsum[4];
  uint32_t red_min[WAVES], red_max[WAVES];
  uint32_t bstart[TS_MAX_RADIX + 1];
};

// One stable counting pass on digit (key >> shift) & mask from `in` to `out` (n entries, whole workgroup of WAVES
// waves; every wave ranks a contiguous chunk, so wave order = input order).
// ON_ID selects the sorted field: the depth bits (.x) or the row id (.y).
template <int WAVES, typename RankT, bool ON_ID>
__device__ __forceinline__ void ts_pass(const uint2* in, uint2* out, RankT* ranks, int n, int shift, uint32_t mask,
                                        TsShared<WAVES>& sh) {
  constexpr int THREADS = WAVES * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int chunk = ((n + THREADS - 1) / THREADS) * 64;  // per-wave contiguous chunk, multiple of 64
  const int cb = wave * chunk, ce = min(n, cb + chunk);
  for (int i = tid; i < WAVES * TS_MAX_RADIX; i += THREADS) (&sh.wave_cnt[0][0])[i] = 0;
  __syncthreads();
  // U items per lane per trip: the U peer masks (nine ballots each) are independent instruction streams; only the
  // running-count update below is serial
  constexpr int U = 1;
  for (int b = cb; b < ce; b += 64 * U) {
    uint32_t d[U], before[U], npeer[U];
    bool valid[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = b + u * 64 + lane;
      valid[u] = i < ce;
      d[u] = valid[u] ? (((ON_ID ? in[i].y : in[i].x) >> shift) & mask) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned long long peers = ts_peers(d[u], valid[u]);
      before[u] = (uint32_t)__popcll(peers & lt_mask);
      npeer[u] = (uint32_t)__popcll(peers);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t prev = sh.wave_cnt[wave][d[u]];
      __builtin_amdgcn_wave_barrier();
      if (valid[u] && before[u] == 0) sh.wave_cnt[wave][d[u]] = prev + npeer[u];
      __builtin_amdgcn_wave_barrier();
      if (valid[u]) ranks[b + u * 64 + lane] = (RankT)(prev + before[u]);
    }
  }
  __syncthreads();
  // the first min(THREADS, 256) threads own DPT consecutive digits each: totals, exclusive scan over the digits,
  // then per-wave start offsets
  constexpr int SCAN_T = THREADS < 256 ? THREADS : 256;
  constexpr int DPT = TS_MAX_RADIX / SCAN_T;
  uint32_t tot[DPT], sum = 0, incl = 0;
  if (tid < SCAN_T) {
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
      tot[j] = 0;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot[j] += sh.wave_cnt[w][DPT * tid + j];
      sum += tot[j];
    }
    incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) sh.wsum[wave] = incl;
  }
  __syncthreads();
  if (tid < SCAN_T) {
    uint32_t run = incl - sum;
#pragma unroll
    for (int w = 0; w < SCAN_T / 64; ++w)
      if (w < wave) run += sh.wsum[w];
#pragma unroll
    for (int j = 0; j < DPT; ++j) {
#pragma unroll
      for (int w = 0; w < WAVES; ++w) {
        const uint32_t c = sh.wave_cnt[w][DPT * tid + j];
        sh.wave_cnt[w][DPT * tid + j] = run;
        run += c;
      }
    }
  }
  __syncthreads();
  for (int b = cb; b < ce; b += 64 * U) {
    uint2 e[U];
    uint32_t dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = b + u * 64 + lane;
      if (i < ce) {
        e[u] = in[i];
        dst[u] = sh.wave_cnt[wave][((ON_ID ? e[u].y : e[u].x) >> shift) & mask] + (uint32_t)ranks[i];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (b + u * 64 + lane < ce) out[dst[u]] = e[u];
  }
  __syncthreads();
}

__device__ __forceinline__ unsigned long long ts_key64(const uint2 e) { return ((unsigned long long)e.x << 32) | e.y; }

template <int WAVES, typename RankT, bool ON_ID>
__device__ __forceinline__ void ts_lsd(uint2*& cur, uint2*& oth, RankT* ranks, int n, int nbits, TsShared<WAVES>& sh) {
  const int npass = (nbits + TS_MAX_BITS - 1) / TS_MAX_BITS;
  const int bpp = npass ? (nbits + npass - 1) / npass : 0;
#pragma unroll 1
  for (int shift = 0; shift < nbits; shift += bpp) {
    const int nb = min(bpp, nbits - shift);
    ts_pass<WAVES, RankT, ON_ID>(cur, oth, ranks, n, shift, (1u << nb) - 1u, sh);
    uint2* t = cur; cur = oth; oth = t;
  }
}

// Sorts n (depth bits, row id) pairs held in `a` (scratch `b`) ascending by (depth bits, row id) -- A.3's order: the
// stable sort of an id-ordered emission.  Returns the buffer that holds the result.  Only the bits that differ
// inside the segment matter (a tile's depths share their high bits).
//   BUCKETS path (LDS only): one unstable counting pass on the top 9 differing depth bits (LDS atomics), then every
//     bucket -- a couple of entries when depths spread -- is insertion-sorted on the full 64-bit (depth, id) key by one
//     lane.  Taken when no bucket exceeds TS_BUCKET_MAX.
//   LSD path: stable 9-bit counting passes over the differing id bits, then over the differing depth bits.
constexpr int TS_BUCKET_MAX = 64;

template <int WAVES, typename RankT, bool BUCKETS>
__device__ __forceinline__ uint2* ts_sort(uint2* a, uint2* b, RankT* ranks, int n, TsShared<WAVES>& sh) {
  constexpr int THREADS = WAVES * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t kmin = 0xffffffffu, kmax = 0u;
  for (int i = tid; i < n; i += THREADS) {
    const uint32_t k = a[i].x;
    kmin = min(kmin, k);
    kmax = max(kmax, k);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));
    kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o));
  }
  if (lane == 0) { sh.red_min[wave] = kmin; sh.red_max[wave] = kmax; }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < WAVES; ++w) {
    kmin = min(kmin, sh.red_min[w]);
    kmax = max(kmax, sh.red_max[w]);
  }
  const uint32_t kdiff = kmin ^ kmax;
  const int nbits = kdiff ? 32 - __clz((int)kdiff) : 0;
  __syncthreads();

  if (BUCKETS) {
    uint32_t* cnt = &sh.wave_cnt[0][0];
    for (int i = tid; i < TS_MAX_RADIX; i += THREADS) cnt[i] = 0;
    __syncthreads();
    // bucket = position of the depth inside the segment's [min, max] on a LINEAR scale (float bit patterns would put
    // half of a frustum's depths into a few exponent buckets); any monotone map of the key keeps the partition valid
    const float dmin = __uint_as_float(kmin);
    const float bscale = (kmax > kmin) ? (float)(TS_MAX_RADIX - 1) / (__uint_as_float(kmax) - dmin) : 0.f;
    auto bucket = [&](uint32_t k) -> uint32_t {
      return (uint32_t)min(TS_MAX_RADIX - 1, (int)((__uint_as_float(k) - dmin) * bscale));
    };
    for (int i = tid; i < n; i += THREADS) atomicAdd(&cnt[bucket(a[i].x)], 1u);
    __syncthreads();
    constexpr int SCAN_T = THREADS < 256 ? THREADS : 256;
    constexpr int DPT = TS_MAX_RADIX / SCAN_T;
    uint32_t tot[DPT], sum = 0, incl = 0, bmax = 0;
    if (tid < SCAN_T) {
#pragma unroll
      for (int j = 0; j < DPT; ++j) {
        tot[j] = cnt[DPT * tid + j];
        sum += tot[j];
        bmax = max(bmax, tot[j]);
      }
      incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) bmax = max(bmax, (uint32_t)__shfl_xor((int)bmax, o));
      if (lane == 63) sh.wsum[wave] = incl;
      if (lane == 0) sh.red_max[wave] = bmax;
    }
    __syncthreads();
    bmax = 0;
#pragma unroll
    for (int w = 0; w < SCAN_T / 64; ++w) bmax = max(bmax, sh.red_max[w]);
    if (bmax <= (uint32_t)TS_BUCKET_MAX) {
      if (tid < SCAN_T) {
        uint32_t run = incl - sum;
#pragma unroll
        for (int w = 0; w < SCAN_T / 64; ++w)
          if (w < wave) run += sh.wsum[w];
#pragma unroll
        for (int j = 0; j < DPT; ++j) {
          sh.bstart[DPT * tid + j] = run;
          cnt[DPT * tid + j] = run;
          run += tot[j];
        }
        if (tid == SCAN_T - 1) sh.bstart[TS_MAX_RADIX] = (uint32_t)n;
      }
      __syncthreads();
      for (int i = tid; i < n; i += THREADS) {
        const uint2 e = a[i];
        b[atomicAdd(&cnt[bucket(e.x)], 1u)] = e;
      }
      __syncthreads();
      for (int d = tid; d < TS_MAX_RADIX; d += THREADS) {
        const int s0 = (int)sh.bstart[d], s1 = (int)sh.bstart[d + 1];
        for (int j = s0 + 1; j < s1; ++j) {
          const uint2 v = b[j];
          const unsigned long long kv = ts_key64(v);
          int k = j - 1;
          while (k >= s0) {
            const uint2 p = b[k];
            if (ts_key64(p) <= kv) break;
            b[k + 1] = p;
            --k;
          }
          b[k + 1] = v;
        }
      }
      __syncthreads();
      return b;
    }
    __syncthreads();
  }

  // LSD: the differing id bits first, then the differing depth bits (every pass stable)
  uint32_t id_or = 0;
  {
    // exact differing-bit mask of the ids across the whole segment
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int i = tid; i < n; i += THREADS) { const uint32_t y = a[i].y; lo = min(lo, y); hi = max(hi, y); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor((int)lo, o));
      hi = max(hi, (uint32_t)__shfl_xor((int)hi, o));
    }
    if (lane == 0) { sh.red_min[wave] = lo; sh.red_max[wave] = hi; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { lo = min(lo, sh.red_min[w]); hi = max(hi, sh.red_max[w]); }
    id_or = lo ^ hi;
    __syncthreads();
  }
  const int ibits = id_or ? 32 - __clz((int)id_or) : 0;
  uint2* cur = a;
  uint2* oth = b;
  ts_lsd<WAVES, RankT, true>(cur, oth, ranks, n, ibits, sh);
  ts_lsd<WAVES, RankT, false>(cur, oth, ranks, n, nbits, sh);
  return cur;
}

// CLS < 0: every tile in grid order, one workgroup each; segments up to CAP in LDS, longer ones on global scratch.
// CLS 0..3: size class CLS of the longest-first order (tile_order, size_classes from gs_raster_tile_order):
//   class 0 = tile_order[0 .. sc[0])  lists > 2048    class 1 = [sc[0] .. sc[1])  1025..2048
//   class 2 = [sc[1] .. sc[2])        513..1024       class 3 = [sc[2] .. n_tiles)  <= 512
// A persistent grid strides over the class.  Class 0 sorts up to CAP in LDS and anything longer on global scratch.
template <int CAP, int CLS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tile_sort_kernel(int32_t T, int32_t tile_n_bits, int64_t n_tiles,
                                                               const int32_t* __restrict__ offsets, int64_t n_isects,
                                                               const int32_t* __restrict__ tile_order,
                                                               const int32_t* __restrict__ size_classes, uint2* pairs,
                                                               int64_t* isect_ids, int32_t* flatten_ids,
                                                               const int32_t* sel, int32_t sel_mode,
                                                               int32_t* sorted_cnt, int32_t* contract_flag,
                                                               const int64_t* __restrict__ n_dev = nullptr) {
  // n_dev: the intersection count in device memory; `n_isects` is then the CAPACITY the host sized the arrays for before
  // it knew the count (gs_isect_bin_build_dev), and a count beyond it means nothing was scattered: do nothing
  if (n_dev) {
    const int64_t n_true = *n_dev;
    if (n_true > n_isects) return;
    n_isects = n_true;
  }
  // sel != NULL restricts the launch to the tiles with sel[tile] < 0 (sel_mode 0: segments the front sort deferred)
  // or sel[tile] != 0 (sel_mode 1: tiles the rasterizer flagged); sorted_cnt[tile] = n marks a segment fully sorted
  constexpr int THREADS = WAVES * 64;
  if (sel && sel_mode == 1 && sel[n_tiles] == 0) return;  // redo launch with no flagged tile (the usual case)
  __shared__ uint2 buf_a[CAP];
  __shared__ uint2 buf_b[CAP];
  __shared__ uint16_t rank_s[CAP];
  __shared__ TsShared<WAVES> sh;
  int64_t k = blockIdx.x, k_end = n_tiles;
  if (CLS >= 0) {
    k += (CLS > 0) ? size_classes[CLS - 1] : 0;
    k_end = (CLS < 3) ? size_classes[CLS] : n_tiles;
  }
  // software pipeline over the workgroup's tiles: the next segment's pairs are loaded into registers while the
  // current one is being sorted in LDS
  constexpr int ITEMS = CAP / THREADS;
  uint2 pre[ITEMS];
  int64_t nx_tile = 0;
  int nx_start = 0, nx_n = 0;
  auto fetch = [&](int64_t kk) {
    nx_tile = (CLS < 0) ? kk : (int64_t)tile_order[kk];
    nx_start = offsets[nx_tile];
    const int e = (nx_tile + 1 < n_tiles) ? offsets[nx_tile + 1] : (int)n_isects;
    nx_n = e - nx_start;
    if (sel && !(sel_mode == 0 ? sel[nx_tile] < 0 : sel[nx_tile] != 0)) nx_n = 0;
    if (nx_n <= CAP) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int i = threadIdx.x + j * THREADS;
        if (i < nx_n) pre[j] = pairs[nx_start + i];
      }
    }
  };
  if (k < k_end) fetch(k);
  for (; k < k_end; k += gridDim.x) {
    const int64_t tile = nx_tile;
    const int start = nx_start, n = nx_n;
    if (n > 0 && n <= CAP) {
#pragma unroll
      for (int j = 0; j < ITEMS; ++j) {
        const int i = threadIdx.x + j * THREADS;
        if (i < n) buf_a[i] = pre[j];
      }
    }
    __syncthreads();
    if (k + gridDim.x < k_end) fetch(k + gridDim.x);
    if (n <= 0) continue;
    const int64_t cam = tile / T, tl = tile - cam * T;
    const int64_t key_hi = (cam << (32 + tile_n_bits)) | (tl << 32);
    if (n <= CAP) {
      const uint2* cur = ts_sort<WAVES, uint16_t, true>(buf_a, buf_b, rank_s, n, sh);
      for (int i = threadIdx.x; i < n; i += THREADS) {
        const uint2 e = cur[i];
        if (isect_ids) isect_ids[start + i] = key_hi | (int64_t)e.x;
        flatten_ids[start + i] = (int32_t)e.y;
      }
    } else if (CLS <= 0 && isect_ids) {  // (without the isect_ids array the host launches no class that needs scratch)
      // very long segment: same algorithm on global scratch -- the pair array, the (not yet written) isect_ids segment
      // as the second pair buffer, the flatten_ids segment as the rank array.  One workgroup owns the whole segment.
      uint2* ga = pairs + start;
      uint2* gb = reinterpret_cast<uint2*>(isect_ids + start);
      uint32_t* gr = reinterpret_cast<uint32_t*>(flatten_ids + start);
      const uint2* cur = ts_sort<WAVES, uint32_t, false>(ga, gb, gr, n, sh);
      // every thread converts its own entries; entry i of either buffer maps onto output slot i
      for (int i = threadIdx.x; i < n; i += THREADS) {
        const uint2 e = cur[i];
        isect_ids[start + i] = key_hi | (int64_t)e.x;
        flatten_ids[start + i] = (int32_t)e.y;
      }
    } else {
      // The list does not fit this class's buffer: tile_order / size_classes do not describe these lists (precondition
      // in the header).  Never leave the segment's ids unwritten -- the rasterizer would gather through whatever the
      // allocation held: hand out the list's own ids unsorted and raise the contract flag.
      for (int i = threadIdx.x; i < n; i += THREADS) {
        const uint2 e = pairs[start + i];
        if (isect_ids) isect_ids[start + i] = key_hi | (int64_t)e.x;
        flatten_ids[start + i] = (int32_t)e.y;
      }
      if (contract_flag && threadIdx.x == 0)
        __hip_atomic_store(contract_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (sorted_cnt && threadIdx.x == 0) sorted_cnt[tile] = n;
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------ front sort
// The compositor stops a tile once every pixel's transmittance falls under 1e-4; on dense scenes that is a few
// hundred entries into a list of thousands (BASELINE configs[4] shape: 7 % of the entries are ever read).  For a
// long segment, sort only its FRONT: histogram the depths into 512 linear buckets (streaming, the segment never has
// to fit in LDS), take the leading buckets that hold at least k_target entries, counting-sort exactly those entries
// into LDS, order every bucket on the 64-bit (depth, id) key and write them out.  The result is the first nf
// entries of the full A.3 order (buckets are depth-monotone, equal depths share a bucket); sorted_cnt[tile] = nf.
// The tail stays unsorted in `pairs` until someone needs it: the rasterizer flags a tile whose pixels are still
// live at the end of the front (gs_isect_tile_sort_redo sorts it fully and the tile is rasterized again), and
// gs_isect_tile_sort completes every tile when the caller asks for the isect_ids / flatten_ids arrays.
// A front that does not fit the LDS buffer or has a crowded bucket is deferred (sorted_cnt = -1) to the full sort.
constexpr int TSF_CAP = 2048;
constexpr int TSF_MIN = 1024;  // segments up to this length are sorted whole by the class 2/3 launches

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void tile_sort_front_kernel(
    int32_t T, int32_t tile_n_bits, int64_t n_tiles, const int32_t* __restrict__ offsets, int64_t n_isects,
    const int32_t* __restrict__ tile_order, const int32_t* __restrict__ size_classes, int32_t k_target,
    const uint2* __restrict__ pairs, int64_t* __restrict__ isect_ids, int32_t* __restrict__ flatten_ids,
    int32_t* __restrict__ sorted_cnt) {
  constexpr int THREADS = WAVES * 64;
  static_assert(THREADS == 256, "the scan below assigns two buckets to each of 256 threads");
  __shared__ uint2 buf[TSF_CAP];
  __shared__ uint32_t cnt[TS_MAX_RADIX];
  __shared__ uint32_t bstart[TS_MAX_RADIX + 1];
  __shared__ uint32_t red_min[WAVES], red_max[WAVES], wsum[WAVES];
  __shared__ int s_bstar, s_bmax[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t k_end = size_classes[1];  // tiles with more than TSF_MIN entries lead the longest-first order
  for (int64_t k = blockIdx.x; k < k_end; k += gridDim.x) {
    const int64_t tile = tile_order[k];
    const int start = offsets[tile];
    const int e_end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
    const int n = e_end - start;
    const uint2* seg = pairs + start;
    // ---- depth range
    uint32_t kmin = 0xffffffffu, kmax = 0u;
    for (int i = tid; i < n; i += THREADS) {
      const uint32_t kk = seg[i].x;
      kmin = min(kmin, kk);
      kmax = max(kmax, kk);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, o));
      kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, o));
    }
    if (lane == 0) { red_min[wave] = kmin; red_max[wave] = kmax; }
    cnt[tid] = 0; cnt[tid + THREADS] = 0;
    if (tid == 0) s_bstar = -1;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < WAVES; ++w) { kmin = min(kmin, red_min[w]); kmax = max(kmax, red_max[w]); }
    const float dmin = __uint_as_float(kmin);
    const float bscale = (kmax > kmin) ? (float)(TS_MAX_RADIX - 1) / (__uint_as_float(kmax) - dmin) : 0.f;
    auto bucket = [&](uint32_t kk) -> int {
      return min(TS_MAX_RADIX - 1, (int)((__uint_as_float(kk) - dmin) * bscale));
    };
    // ---- histogram, exclusive scan (two buckets per thread)
    for (int i = tid; i < n; i += THREADS) atomicAdd(&cnt[bucket(seg[i].x)], 1u);
    __syncthreads();
    const uint32_t t0 = cnt[2 * tid], t1 = cnt[2 * tid + 1];
    uint32_t incl = t0 + t1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - (t0 + t1);
#pragma unroll
    for (int w = 0; w < WAVES; ++w)
      if (w < wave) run += wsum[w];
    bstart[2 * tid] = run;
    bstart[2 * tid + 1] = run + t0;
    if (tid == THREADS - 1) bstart[TS_MAX_RADIX] = (uint32_t)n;
    // the bucket in which the running count reaches k_target closes the front
    if (run < (uint32_t)k_target && run + t0 >= (uint32_t)k_target) s_bstar = 2 * tid;
    else if (run + t0 < (uint32_t)k_target && run + t0 + t1 >= (uint32_t)k_target) s_bstar = 2 * tid + 1;
    __syncthreads();
    const int bstar = s_bstar;
    const int nf = bstar >= 0 ? (int)bstart[bstar + 1] : n + TSF_CAP;
    // crowded bucket inside the front?  (one lane orders a bucket by insertion)
    int bmx = 0;
    if (2 * tid <= bstar) bmx = (int)t0;
    if (2 * tid + 1 <= bstar) bmx = max(bmx, (int)t1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bmx = max(bmx, __shfl_xor(bmx, o));
    if (lane == 0) s_bmax[wave] = bmx;
    __syncthreads();
    bmx = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) bmx = max(bmx, s_bmax[w]);
    if (nf > TSF_CAP || bmx > TS_BUCKET_MAX) {
      if (tid == 0) sorted_cnt[tile] = -1;
      __syncthreads();
      continue;
    }
    // ---- counting placement of the front entries, then per-bucket insertion sort on (depth, id)
    if (2 * tid <= bstar) cnt[2 * tid] = bstart[2 * tid];
    if (2 * tid + 1 <= bstar) cnt[2 * tid + 1] = bstart[2 * tid + 1];
    __syncthreads();
    for (int i = tid; i < n; i += THREADS) {
      const uint2 en = seg[i];
      const int b = bucket(en.x);
      if (b <= bstar) buf[atomicAdd(&cnt[b], 1u)] = en;
    }
    __syncthreads();
    for (int d = tid; d <= bstar; d += THREADS) {
      const int s0 = (int)bstart[d], s1 = (int)bstart[d + 1];
      for (int j = s0 + 1; j < s1; ++j) {
        const uint2 v = buf[j];
        const unsigned long long kv = ts_key64(v);
        int q = j - 1;
        while (q >= s0) {
          const uint2 pv = buf[q];
          if (ts_key64(pv) <= kv) break;
          buf[q + 1] = pv;
          --q;
        }
        buf[q + 1] = v;
      }
    }
    __syncthreads();
    const int64_t cam = tile / T, tl = tile - cam * T;
    const int64_t key_hi = (cam << (32 + tile_n_bits)) | (tl << 32);
    for (int i = tid; i < nf; i += THREADS) {
      const uint2 en = buf[i];
      if (isect_ids) isect_ids[start + i] = key_hi | (int64_t)en.x;
      flatten_ids[start + i] = (int32_t)en.y;
    }
    if (tid == 0) sorted_cnt[tile] = nf;
    __syncthreads();
  }
}

static inline int bin_ilog2_floor(int64_t v) {
  int b = 0;
  while (v >>= 1) ++b;
  return b;
}

}  // namespace gs

// launch shapes of the per-tile sort's size classes (waves per list, grid multiples of 256) and the LDS budget of the
// staged region scatter; each value is the best of a sweep on the box (profiles/r02_ab_runs.txt)
#define GS_TS_W1 8   // 1025..2048 entries: few lists, each a latency chain
#define GS_TS_W2 4   // 513..1024
#define GS_TS_W3 2   // <= 512
#define GS_TS_G2 16
#define GS_TS_G3 14
#define GS_SCATTER_LDS (163840 - 256)  // cursors + staging area, one workgroup per CU
static int bin_check(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices) {
  if (C < 1 || N < 0 || tile_w < 1 || tile_h < 1 || n_slices < 1) return GS_ERR_INVALID_ARG;
  if ((int64_t)tile_w * tile_h > gs::BIN_MAX_TILES) return GS_ERR_UNSUPPORTED;
  if ((int64_t)C * N >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (C > 65535 || n_slices > 65535) return GS_ERR_UNSUPPORTED;
  return GS_OK;
}

extern "C" int32_t gs_isect_bin_max_tiles(void) { return gs::BIN_MAX_TILES; }

static int bin_slice_rows(int32_t N, int32_t n_slices) {
  int rows = (int)gs::ceil_div(N > 0 ? N : 1, n_slices);
  return (int)gs::ceil_div(rows, gs::BIN_THREADS) * gs::BIN_THREADS;
}

extern "C" int32_t gs_isect_bin_slice_rows(int32_t N, int32_t n_slices) {
  return n_slices < 1 ? 0 : bin_slice_rows(N, n_slices);
}

extern "C" int gs_isect_bin_count(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                                  const int32_t* tiles_per_gauss, int32_t n_slices, uint32_t* table,
                                  uint64_t* vis_list, int32_t* vis_counts, uint64_t* vis_masks, int32_t tile_size,
                                  void* stream) {
  int st = bin_check(C, N, tile_w, tile_h, n_slices);
  if (st != GS_OK) return st;
  if (!table || (N > 0 && (!splats || !tiles_per_gauss))) return GS_ERR_INVALID_ARG;
  if ((vis_list == nullptr) != (vis_counts == nullptr) || ((uintptr_t)vis_list & 15)) return GS_ERR_INVALID_ARG;
  if (vis_masks && (!vis_list || tile_size < 1 || ((uintptr_t)vis_masks & 7))) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  const dim3 grid((unsigned)n_slices, (unsigned)C);
  const size_t lds = (size_t)T * sizeof(uint32_t);
  const int rows = bin_slice_rows(N, n_slices);
  if (vis_masks)
    hipLaunchKernelGGL((gs::isect_bin_kernel<false, true>), grid, dim3(gs::BIN_THREADS), lds, (hipStream_t)stream, N,
                       tile_w, T, n_slices, rows, splats, tiles_per_gauss, table, (const int32_t*)nullptr,
                       (uint2*)nullptr, reinterpret_cast<uint4*>(vis_list), vis_counts,
                       reinterpret_cast<uint2*>(vis_masks), (float)tile_size);
  else
    hipLaunchKernelGGL((gs::isect_bin_kernel<false, false>), grid, dim3(gs::BIN_THREADS), lds, (hipStream_t)stream, N,
                       tile_w, T, n_slices, rows, splats, tiles_per_gauss, table, (const int32_t*)nullptr,
                       (uint2*)nullptr, reinterpret_cast<uint4*>(vis_list), vis_counts, (uint2*)nullptr, 0.f);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_bin_offsets(int32_t C, int32_t tile_w, int32_t tile_h, int32_t n_slices, uint32_t* table,
                                    int32_t* tile_counts, int32_t* isect_offsets, int64_t* total, void* stream) {
  return gs_isect_bin_offsets_post(C, tile_w, tile_h, n_slices, table, tile_counts, isect_offsets, total, nullptr,
                                   nullptr, stream);
}

extern "C" int gs_isect_bin_offsets_post(int32_t C, int32_t tile_w, int32_t tile_h, int32_t n_slices, uint32_t* table,
                                         int32_t* tile_counts, int32_t* isect_offsets, int64_t* total,
                                         int64_t* total_post, int32_t* size_classes_post, void* stream) {
  if (C < 1 || tile_w < 1 || tile_h < 1 || n_slices < 1) return GS_ERR_INVALID_ARG;
  if (!table || !tile_counts || !isect_offsets || !total) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  const int64_t n_tiles = (int64_t)C * T;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs::bin_slice_prefix_kernel, dim3((unsigned)gs::ceil_div(n_tiles, 64)), dim3(256), 0, s, C, T,
                     n_slices, table, tile_counts);
#define GS_BO_ARGS dim3(1), dim3(1024), 0, s, n_tiles, tile_counts, isect_offsets, total, total_post, size_classes_post
  if (n_tiles <= 1024 * 4)
    hipLaunchKernelGGL(gs::bin_offsets_kernel<4>, GS_BO_ARGS);
  else if (n_tiles <= 1024 * 10)
    hipLaunchKernelGGL(gs::bin_offsets_kernel<10>, GS_BO_ARGS);
  else
    hipLaunchKernelGGL(gs::bin_offsets_kernel<16>, GS_BO_ARGS);
#undef GS_BO_ARGS
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_bin_scatter(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                                    const int32_t* tiles_per_gauss, int32_t n_slices, const uint32_t* table,
                                    const int32_t* isect_offsets, uint64_t* pairs, void* stream) {
  int st = bin_check(C, N, tile_w, tile_h, n_slices);
  if (st != GS_OK) return st;
  if (N == 0) return GS_OK;
  if (!splats || !tiles_per_gauss || !table || !isect_offsets || !pairs) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  hipLaunchKernelGGL((gs::isect_bin_kernel<true, false>), dim3((unsigned)n_slices, (unsigned)C), dim3(gs::BIN_THREADS),
                     (size_t)T * sizeof(uint32_t), (hipStream_t)stream, N, tile_w, T, n_slices,
                     bin_slice_rows(N, n_slices), splats, tiles_per_gauss, const_cast<uint32_t*>(table), isect_offsets,
                     reinterpret_cast<uint2*>(pairs), (uint4*)nullptr, (int32_t*)nullptr, (uint2*)nullptr, 0.f);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

static int scatter_regions_impl(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                                int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                                const int32_t* vis_counts, const uint64_t* vis_masks, const uint32_t* table,
                                const int32_t* isect_offsets, const int32_t* tile_counts, uint64_t* pairs,
                                const int64_t* n_dev, int64_t pair_cap, void* stream) {
  int st = bin_check(C, N, tile_w, tile_h, n_slices);
  if (st != GS_OK) return st;
  if (slices_per_group < 1 || n_regions < 1 || n_regions > 65535) return GS_ERR_INVALID_ARG;
  if (N == 0) return GS_OK;
  if (!vis_list || !vis_counts || !table || !isect_offsets || !pairs || ((uintptr_t)vis_list & 15) ||
      ((uintptr_t)vis_masks & 7))
    return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  const int groups = (int)gs::ceil_div(n_slices, slices_per_group);
  if ((int64_t)groups * n_regions > 0x7fffffff) return GS_ERR_UNSUPPORTED;
  const int band_rows = (int)gs::ceil_div(tile_h, n_regions) + 1;
  const dim3 grid((unsigned)(groups * n_regions), (unsigned)C);
  const size_t lds = (size_t)band_rows * tile_w * sizeof(uint32_t);
  // staged through LDS when the list lengths are at hand and the band's cursors leave room for a useful staging area
  const int64_t nt_max = (int64_t)band_rows * tile_w;
  const int64_t cap = ((GS_SCATTER_LDS - 8 * nt_max) / 10) & ~(int64_t)63;
  if (tile_counts && cap >= 2048) {
    const size_t lds_staged = (size_t)(8 * nt_max + 10 * cap);
    const void* fn = vis_masks ? (const void*)gs::isect_region_scatter_staged_kernel<true>
                               : (const void*)gs::isect_region_scatter_staged_kernel<false>;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_staged);
    if (e != hipSuccess) return (int)e;
    if (vis_masks)
      hipLaunchKernelGGL(gs::isect_region_scatter_staged_kernel<true>, grid, dim3(gs::BIN_THREADS), lds_staged,
                         (hipStream_t)stream, tile_w, tile_h, T, n_slices, slices_per_group, n_regions,
                         bin_slice_rows(N, n_slices), (int32_t)nt_max, (int32_t)cap,
                         reinterpret_cast<const uint4*>(vis_list), vis_counts,
                         reinterpret_cast<const uint2*>(vis_masks), table, isect_offsets, tile_counts,
                         reinterpret_cast<uint2*>(pairs), n_dev, pair_cap);
    else
      hipLaunchKernelGGL(gs::isect_region_scatter_staged_kernel<false>, grid, dim3(gs::BIN_THREADS), lds_staged,
                         (hipStream_t)stream, tile_w, tile_h, T, n_slices, slices_per_group, n_regions,
                         bin_slice_rows(N, n_slices), (int32_t)nt_max, (int32_t)cap,
                         reinterpret_cast<const uint4*>(vis_list), vis_counts, (const uint2*)nullptr, table,
                         isect_offsets, tile_counts, reinterpret_cast<uint2*>(pairs), n_dev, pair_cap);
    GS_CHECK_LAUNCH();
    return GS_OK;
  }
  if (vis_masks)
    hipLaunchKernelGGL(gs::isect_region_scatter_kernel<true>, grid, dim3(gs::BIN_THREADS), lds, (hipStream_t)stream,
                       tile_w, tile_h, T, n_slices, slices_per_group, n_regions, bin_slice_rows(N, n_slices),
                       reinterpret_cast<const uint4*>(vis_list), vis_counts, reinterpret_cast<const uint2*>(vis_masks),
                       table, isect_offsets, reinterpret_cast<uint2*>(pairs), n_dev, pair_cap);
  else
    hipLaunchKernelGGL(gs::isect_region_scatter_kernel<false>, grid, dim3(gs::BIN_THREADS), lds, (hipStream_t)stream,
                       tile_w, tile_h, T, n_slices, slices_per_group, n_regions, bin_slice_rows(N, n_slices),
                       reinterpret_cast<const uint4*>(vis_list), vis_counts, (const uint2*)nullptr, table,
                       isect_offsets, reinterpret_cast<uint2*>(pairs), n_dev, pair_cap);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_bin_scatter_regions(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                                            int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                                            const int32_t* vis_counts, const uint64_t* vis_masks,
                                            const uint32_t* table, const int32_t* isect_offsets,
                                            const int32_t* tile_counts, uint64_t* pairs, void* stream) {
  return scatter_regions_impl(C, N, tile_w, tile_h, n_slices, slices_per_group, n_regions, vis_list, vis_counts,
                              vis_masks, table, isect_offsets, tile_counts, pairs, nullptr, 0, stream);
}

// `sch` (host copy of size_classes, or NULL): lets the host skip the launches of size classes that hold no tile -- on the
// shapes the reference trains at only the shortest class exists, and a launch that finds nothing to do still costs a few
// microseconds of a launch-bound step.  Class c of the longest-first order is [sch[c-1], sch[c]) with sch[-1] = 0 and
// sch[3] = n_tiles.
static inline bool ts_class_empty(const int32_t* sch, int cls, int64_t n_tiles) {
  if (!sch) return false;
  const int64_t lo = cls > 0 ? sch[cls - 1] : 0, hi = cls < 3 ? sch[cls] : n_tiles;
  return hi <= lo;
}

// isect_ids may be NULL (a caller that only rasterizes needs flatten_ids and the offsets) when no list is longer than
// 2048 entries -- the longest class sorts on global scratch and borrows the isect_ids segment as its second buffer --
// which the host can only know from its copy of the size classes.
static inline bool ts_ids_optional(const int32_t* tile_order, const int32_t* size_classes, const int32_t* sch,
                                   int64_t n_tiles) {
  return tile_order && size_classes && sch && ts_class_empty(sch, 0, n_tiles);
}

static int tile_sort_impl(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets, int64_t n_isects,
                          const int32_t* tile_order, const int32_t* size_classes, uint64_t* pairs, int64_t* isect_ids,
                          int32_t* flatten_ids, const int32_t* sch, void* stream, const int64_t* n_dev = nullptr) {
  if (C < 1 || tile_w < 1 || tile_h < 1 || n_isects < 0) return GS_ERR_INVALID_ARG;
  if (gs_contract_take()) return GS_ERR_CONTRACT;
  if (n_isects == 0) return GS_OK;
  if (n_isects >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (!isect_offsets || !pairs || !flatten_ids) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  // (n_dev: the host does not know the size classes yet; a list beyond the longest class's LDS buffer without the
  // isect_ids scratch raises the contract flag, which the caller of gs_isect_bin_build_dev checks for)
  if (!isect_ids && !n_dev && !ts_ids_optional(tile_order, size_classes, sch, (int64_t)C * T)) return GS_ERR_INVALID_ARG;
  const int tile_n_bits = gs::bin_ilog2_floor(T) + 1;
  const int cam_n_bits = gs::bin_ilog2_floor(C) + 1;
  if (32 + tile_n_bits + cam_n_bits > 64) return GS_ERR_UNSUPPORTED;
  const int64_t n_tiles = (int64_t)C * T;
  hipStream_t s = (hipStream_t)stream;
  uint2* pr = reinterpret_cast<uint2*>(pairs);
  int32_t* cf = gs_contract_flag_device();
#define GS_TS_ARGS T, tile_n_bits, n_tiles, isect_offsets, n_isects, tile_order, size_classes, pr, isect_ids, flatten_ids, (const int32_t*)nullptr, 0, (int32_t*)nullptr, cf, n_dev
  if (tile_order && size_classes) {
    // longest lists first; grids are persistent and sized to the LDS-limited residency of each class; the short
    // classes run 4 / 2 waves per segment (one wave each left 8 waves per CU: 0.136 -> 0.110 ms for the chain)
    if (!ts_class_empty(sch, 0, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<6144, 0, 16>), dim3(256), dim3(1024), 0, s, GS_TS_ARGS);
    if (!ts_class_empty(sch, 1, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<2048, 1, GS_TS_W1>), dim3(256 * 3), dim3(64 * GS_TS_W1), 0, s, GS_TS_ARGS);
    if (!ts_class_empty(sch, 2, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<1024, 2, GS_TS_W2>), dim3(256 * GS_TS_G2), dim3(64 * GS_TS_W2), 0, s, GS_TS_ARGS);
    if (!ts_class_empty(sch, 3, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<512, 3, GS_TS_W3>), dim3(256 * GS_TS_G3), dim3(64 * GS_TS_W3), 0, s, GS_TS_ARGS);
  } else {
    hipLaunchKernelGGL((gs::tile_sort_kernel<2048, -1, 4>), dim3((unsigned)n_tiles), dim3(256), 0, s, GS_TS_ARGS);
  }
#undef GS_TS_ARGS
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_tile_sort(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                  int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes, uint64_t* pairs,
                                  int64_t* isect_ids, int32_t* flatten_ids, void* stream) {
  return tile_sort_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, pairs, isect_ids,
                        flatten_ids, nullptr, stream);
}

// gs_isect_bin_scatter_regions + gs_isect_tile_sort for a host that has NOT read n_isects yet (include/gsplat_mi355x.h,
// "device-side n_isects"): `capacity` sizes pairs / flatten_ids (/ isect_ids), the count is read from *n_isects_dev by
// the kernels; all four size classes are launched (the host does not know which ones hold tiles) and every list is sorted
// whole.  A count beyond the capacity leaves the arrays untouched.
extern "C" int gs_isect_bin_build_dev(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices,
                                      int32_t slices_per_group, int32_t n_regions, const uint64_t* vis_list,
                                      const int32_t* vis_counts, const uint64_t* vis_masks, const uint32_t* table,
                                      const int32_t* isect_offsets, int64_t capacity, const int64_t* n_isects_dev,
                                      const int32_t* tile_order, const int32_t* size_classes, uint64_t* pairs,
                                      int64_t* isect_ids, int32_t* flatten_ids, const int32_t* tile_counts,
                                      void* stream) {
  if (!n_isects_dev || capacity < 1 || !tile_order || !size_classes) return GS_ERR_INVALID_ARG;
  int st = scatter_regions_impl(C, N, tile_w, tile_h, n_slices, slices_per_group, n_regions, vis_list, vis_counts,
                                vis_masks, table, isect_offsets, tile_counts, pairs, n_isects_dev, capacity, stream);
  if (st != GS_OK) return st;
  if (N == 0) return GS_OK;
  return tile_sort_impl(C, tile_w, tile_h, isect_offsets, capacity, tile_order, size_classes, pairs, isect_ids,
                        flatten_ids, nullptr, stream, n_isects_dev);
}

extern "C" int gs_isect_tile_sort_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                     int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                     uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                                     const int32_t* size_classes_host, void* stream) {
  return tile_sort_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, pairs, isect_ids,
                        flatten_ids, size_classes_host, stream);
}

static int tile_sort_geom(int32_t C, int32_t tile_w, int32_t tile_h, int64_t n_isects, int* T, int* tile_n_bits) {
  if (C < 1 || tile_w < 1 || tile_h < 1 || n_isects < 0) return GS_ERR_INVALID_ARG;
  if (n_isects >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  *T = tile_w * tile_h;
  *tile_n_bits = gs::bin_ilog2_floor(*T) + 1;
  if (32 + *tile_n_bits + gs::bin_ilog2_floor(C) + 1 > 64) return GS_ERR_UNSUPPORTED;
  if (gs_contract_take()) return GS_ERR_CONTRACT;
  return GS_OK;
}

static int tile_sort_front_impl(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                int32_t k_target, uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                                int32_t* sorted_counts, const int32_t* sch, void* stream) {
  int T, tile_n_bits;
  int st = tile_sort_geom(C, tile_w, tile_h, n_isects, &T, &tile_n_bits);
  if (st != GS_OK) return st;
  if (n_isects == 0) return GS_OK;
  if (!isect_offsets || !tile_order || !size_classes || !pairs || !flatten_ids || !sorted_counts)
    return GS_ERR_INVALID_ARG;
  if (k_target < 1 || k_target > gs::TSF_MIN) return GS_ERR_INVALID_ARG;
  const int64_t n_tiles = (int64_t)C * T;
  if (!isect_ids && !ts_ids_optional(tile_order, size_classes, sch, n_tiles)) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  uint2* pr = reinterpret_cast<uint2*>(pairs);
  int32_t* cf = gs_contract_flag_device();
#define GS_TSF_ARGS T, tile_n_bits, n_tiles, isect_offsets, n_isects, tile_order, size_classes, pr, isect_ids, flatten_ids
  if (!(ts_class_empty(sch, 0, n_tiles) && ts_class_empty(sch, 1, n_tiles))) {  // any tile with more than TSF_MIN entries
    hipLaunchKernelGGL((gs::tile_sort_front_kernel<4>), dim3(256 * 6), dim3(256), 0, s, T, tile_n_bits, n_tiles,
                       isect_offsets, n_isects, tile_order, size_classes, k_target, (const uint2*)pr, isect_ids,
                       flatten_ids, sorted_counts);
    // segments the front kernel deferred (sorted_counts < 0): whole-segment sort, as gs_isect_tile_sort does
    if (!ts_class_empty(sch, 0, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<6144, 0, 16>), dim3(256), dim3(1024), 0, s, GS_TSF_ARGS,
                         (const int32_t*)sorted_counts, 0, sorted_counts, cf);
    if (!ts_class_empty(sch, 1, n_tiles))
      hipLaunchKernelGGL((gs::tile_sort_kernel<2048, 1, GS_TS_W1>), dim3(256 * 3), dim3(64 * GS_TS_W1), 0, s, GS_TSF_ARGS,
                         (const int32_t*)sorted_counts, 0, sorted_counts, cf);
  }
  // short segments: sorted whole (sorted_counts was pre-filled with the list lengths by the caller)
  if (!ts_class_empty(sch, 2, n_tiles))
    hipLaunchKernelGGL((gs::tile_sort_kernel<1024, 2, GS_TS_W2>), dim3(256 * GS_TS_G2), dim3(64 * GS_TS_W2), 0, s,
                       GS_TSF_ARGS, (const int32_t*)nullptr, 0, (int32_t*)nullptr, cf);
  if (!ts_class_empty(sch, 3, n_tiles))
    hipLaunchKernelGGL((gs::tile_sort_kernel<512, 3, GS_TS_W3>), dim3(256 * GS_TS_G3), dim3(64 * GS_TS_W3), 0, s,
                       GS_TSF_ARGS, (const int32_t*)nullptr, 0, (int32_t*)nullptr, cf);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_tile_sort_front(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                        int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                        int32_t k_target, uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                                        int32_t* sorted_counts, void* stream) {
  return tile_sort_front_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, k_target, pairs,
                              isect_ids, flatten_ids, sorted_counts, nullptr, stream);
}

extern "C" int gs_isect_tile_sort_front_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                           int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                           int32_t k_target, uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                                           int32_t* sorted_counts, const int32_t* size_classes_host, void* stream) {
  return tile_sort_front_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, k_target, pairs,
                              isect_ids, flatten_ids, sorted_counts, size_classes_host, stream);
}

static int tile_sort_redo_impl(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                               int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                               const int32_t* redo_flags, uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                               int32_t* sorted_counts, const int32_t* sch, void* stream) {
  int T, tile_n_bits;
  int st = tile_sort_geom(C, tile_w, tile_h, n_isects, &T, &tile_n_bits);
  if (st != GS_OK) return st;
  if (n_isects == 0) return GS_OK;
  if (!isect_offsets || !tile_order || !size_classes || !redo_flags || !pairs || !flatten_ids || !sorted_counts)
    return GS_ERR_INVALID_ARG;
  const int64_t n_tiles = (int64_t)C * T;
  if (!isect_ids && !ts_ids_optional(tile_order, size_classes, sch, n_tiles)) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  uint2* pr = reinterpret_cast<uint2*>(pairs);
  int32_t* cf = gs_contract_flag_device();
  if (!ts_class_empty(sch, 0, n_tiles))
    hipLaunchKernelGGL((gs::tile_sort_kernel<6144, 0, 16>), dim3(256), dim3(1024), 0, s, GS_TSF_ARGS, redo_flags, 1,
                       sorted_counts, cf);
  if (!ts_class_empty(sch, 1, n_tiles))
    hipLaunchKernelGGL((gs::tile_sort_kernel<2048, 1, GS_TS_W1>), dim3(256 * 3), dim3(64 * GS_TS_W1), 0, s, GS_TSF_ARGS, redo_flags, 1,
                       sorted_counts, cf);
#undef GS_TSF_ARGS
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_tile_sort_redo(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                       int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                       const int32_t* redo_flags, uint64_t* pairs, int64_t* isect_ids,
                                       int32_t* flatten_ids, int32_t* sorted_counts, void* stream) {
  return tile_sort_redo_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, redo_flags, pairs,
                             isect_ids, flatten_ids, sorted_counts, nullptr, stream);
}

extern "C" int gs_isect_tile_sort_redo_ex(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                          int64_t n_isects, const int32_t* tile_order, const int32_t* size_classes,
                                          const int32_t* redo_flags, uint64_t* pairs, int64_t* isect_ids,
                                          int32_t* flatten_ids, int32_t* sorted_counts,
                                          const int32_t* size_classes_host, void* stream) {
  return tile_sort_redo_impl(C, tile_w, tile_h, isect_offsets, n_isects, tile_order, size_classes, redo_flags, pairs,
                             isect_ids, flatten_ids, sorted_counts, size_classes_host, stream);
}
