// isect_tiles + sort + isect_offset_encode as ONE binned pipeline for gfx950.
//
// gsplat emits one 64-bit (camera|tile|depth) key per intersection and radix-sorts all of them (6 sweeps of a
// 12-B/element array at config 2).  The result is fully determined by three facts: which (row, tile) pairs exist,
// the row depths, and the tie rule (equal keys keep ascending flat row id).  So build it directly:
//   1. gs_isect_bin_count    every workgroup owns one camera and a slice of its rows and counts tile hits in an LDS
//                            histogram over that camera's tiles  ->  table[camera][slice][tile]
//   2. gs_isect_bin_offsets  per tile: exclusive prefix over the slices, total per tile; exclusive scan of the totals
//                            = isect_offsets (A.4's lower bounds, no search needed), last value = n_isects
//   3. gs_isect_bin_scatter  same walk; an LDS cursor per tile (offset + slice prefix) hands every hit its slot in the
//                            tile's segment; writes (depth bits, flat row id) pairs, order inside a segment arbitrary
//   4. gs_isect_tile_sort    one workgroup per tile: stable LSD radix sort of the segment by depth IN LDS (4 x 8 bits,
//                            wave64 match-any ranking), then an odd-even fix-up that orders equal depths by row id,
//                            then writes the final isect_ids / flatten_ids.  Segments beyond the LDS capacity run the
//                            same code on global scratch.
// Bit-exact with A.3/A.4 (same keys, same order, same offsets); the atomics only decide the order INSIDE a tile
// segment before step 4, which the sort then fixes completely.  HBM traffic: 2 reads of the visible records,
// one 8-B write + read and one 12-B write per intersection.
#include "gs_common.hpp"

namespace gs {

constexpr int BIN_THREADS = 256;
constexpr int BIN_COOP = 16;            // tile boxes larger than this are walked by the whole wave
constexpr int BIN_MAX_TILES = 16384;    // per camera: LDS histogram of 64 KiB

// ------------------------------------------------------------------------------------------ count / scatter
template <bool SCATTER>
__global__ __launch_bounds__(BIN_THREADS) void isect_bin_kernel(int32_t N, int32_t tile_w, int32_t T, int32_t n_slices,
                                                                int32_t rows_per_slice,
                                                                const float* __restrict__ splats,
                                                                const int32_t* __restrict__ tiles_per_gauss,
                                                                uint32_t* __restrict__ table,
                                                                const int32_t* __restrict__ offsets,
                                                                uint2* __restrict__ pairs) {
  extern __shared__ uint32_t cnt[];  // [T]
  const int c = blockIdx.y, slice = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63;
  uint32_t* trow = table + ((int64_t)c * n_slices + slice) * T;
  for (int t = tid; t < T; t += BIN_THREADS) cnt[t] = SCATTER ? (uint32_t)offsets[(int64_t)c * T + t] + trow[t] : 0u;
  __syncthreads();
  const int n0 = slice * rows_per_slice;
  const int n1 = min(N, n0 + rows_per_slice);
  for (int nb = n0; nb < n1; nb += BIN_THREADS) {
    const int n = nb + tid;
    int cntg = 0, x0 = 0, y0 = 0, x1 = 0;
    uint32_t dbits = 0;
    const int64_t idx = (int64_t)c * N + n;
    if (n < n1) {
      cntg = tiles_per_gauss[idx];
      if (cntg > 0) {
        const float4 r3 = reinterpret_cast<const float4*>(splats + idx * REC)[3];
        const int mn = __float_as_int(r3.y), mx = __float_as_int(r3.z);
        x0 = mn & 0xffff; y0 = mn >> 16; x1 = mx & 0xffff;
        if (SCATTER) dbits = (uint32_t)__float_as_int(splats[idx * REC + 10]);
      }
    }
    if (cntg > 0 && cntg <= BIN_COOP) {
      int tj = x0, trow0 = y0 * tile_w;
      for (int t = 0; t < cntg; ++t) {
        const uint32_t pos = atomicAdd(&cnt[trow0 + tj], 1u);
        if (SCATTER) pairs[pos] = make_uint2(dbits, (uint32_t)idx);
        if (++tj == x1) { tj = x0; trow0 += tile_w; }
      }
    }
    unsigned long long big = __ballot(cntg > BIN_COOP);
    while (big) {
      const int src = __ffsll((long long)big) - 1;
      big &= big - 1;
      const int b_cnt = __shfl(cntg, src), b_x0 = __shfl(x0, src), b_y0 = __shfl(y0, src);
      const int b_w = __shfl(x1, src) - b_x0;
      const uint32_t b_d = __shfl(dbits, src);
      const uint32_t b_idx = (uint32_t)(idx - lane + src);
      for (int t = lane; t < b_cnt; t += 64) {
        const int ti = b_y0 + t / b_w, tj = b_x0 + t % b_w;
        const uint32_t pos = atomicAdd(&cnt[ti * tile_w + tj], 1u);
        if (SCATTER) pairs[pos] = make_uint2(b_d, b_idx);
      }
    }
  }
  if (!SCATTER) {
    __syncthreads();
    for (int t = tid; t < T; t += BIN_THREADS) trow[t] = cnt[t];
  }
}

// per tile: exclusive prefix over the slices (in place), total -> counts
__global__ __launch_bounds__(256) void bin_slice_prefix_kernel(int32_t C, int32_t T, int32_t n_slices,
                                                               uint32_t* __restrict__ table,
                                                               int32_t* __restrict__ counts) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (int64_t)C * T) return;
  const int c = (int)(g / T), t = (int)(g - (int64_t)c * T);
  uint32_t* col = table + (int64_t)c * n_slices * T + t;
  uint32_t run = 0;
  int s = 0;
  for (; s + 4 <= n_slices; s += 4) {
    const uint32_t v0 = col[(int64_t)(s + 0) * T], v1 = col[(int64_t)(s + 1) * T];
    const uint32_t v2 = col[(int64_t)(s + 2) * T], v3 = col[(int64_t)(s + 3) * T];
    col[(int64_t)(s + 0) * T] = run; run += v0;
    col[(int64_t)(s + 1) * T] = run; run += v1;
    col[(int64_t)(s + 2) * T] = run; run += v2;
    col[(int64_t)(s + 3) * T] = run; run += v3;
  }
  for (; s < n_slices; ++s) {
    const uint32_t v = col[(int64_t)s * T];
    col[(int64_t)s * T] = run;
    run += v;
  }
  counts[g] = (int32_t)run;
}

// single workgroup: exclusive scan of the per-tile totals -> offsets, grand total -> *total
__global__ __launch_bounds__(1024) void bin_offsets_kernel(int64_t n_tiles, const int32_t* __restrict__ counts,
                                                           int32_t* __restrict__ offsets, int64_t* __restrict__ total) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  constexpr int ITEMS = 4;
  for (int64_t base = 0; base < n_tiles; base += 1024 * ITEMS) {
    const int64_t i0 = base + (int64_t)tid * ITEMS;
    int32_t v[ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
      v[i] = (i0 + i < n_tiles) ? counts[i0 + i] : 0;
      s += v[i];
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
  if (tid == 0) *total = carry_s;
}

// ------------------------------------------------------------------------------------------ per-tile sort
constexpr int TS_THREADS = 256;
constexpr int TS_WAVES = 4;
constexpr int TS_CAP = 2048;  // segment entries sorted in LDS; longer segments use global scratch

// lanes (among the valid ones) holding the same 8-bit digit
__device__ __forceinline__ unsigned long long ts_peers(uint32_t d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < 8; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// One stable counting pass on digit (key >> shift) & 255 from `in` to `out` (n entries, whole workgroup).
// Returns false (and moves nothing) when every key has the same digit.
template <typename RankT>
__device__ __forceinline__ bool ts_pass(const uint2* in, uint2* out, RankT* ranks, int n, int shift,
                                        uint32_t (*wave_cnt)[256], uint32_t* wsum, int* flag) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int chunk = ((n + TS_WAVES * 64 - 1) / (TS_WAVES * 64)) * 64;  // per-wave contiguous chunk, multiple of 64
  const int cb = wave * chunk, ce = min(n, cb + chunk);
#pragma unroll
  for (int w = 0; w < TS_WAVES; ++w) wave_cnt[w][tid] = 0;
  if (tid == 0) *flag = 0;
  __syncthreads();
  for (int b = cb; b < ce; b += 64) {
    const int i = b + lane;
    const bool valid = i < ce;
    const uint32_t d = valid ? ((in[i].x >> shift) & 255u) : 0u;
    const unsigned long long peers = ts_peers(d, valid);
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t prev = wave_cnt[wave][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wave_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    if (valid) ranks[i] = (RankT)(prev + before);
  }
  __syncthreads();
  {
    // thread d: digit total, exclusive scan over the digits, then per-wave start offsets
    uint32_t tot = 0;
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) tot += wave_cnt[w][tid];
    if (tot == (uint32_t)n) *flag = 1;
    uint32_t incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - tot;
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w)
      if (w < wave) run += wsum[w];
#pragma unroll
    for (int w = 0; w < TS_WAVES; ++w) {
      const uint32_t c = wave_cnt[w][tid];
      wave_cnt[w][tid] = run;
      run += c;
    }
  }
  __syncthreads();
  if (*flag) return false;
  for (int b = cb; b < ce; b += 64) {
    const int i = b + lane;
    if (i < ce) {
      const uint2 e = in[i];
      out[wave_cnt[wave][(e.x >> shift) & 255u] + (uint32_t)ranks[i]] = e;
    }
  }
  __syncthreads();
  return true;
}

template <typename RankT>
__device__ __forceinline__ uint2* ts_sort(uint2* a, uint2* b, RankT* ranks, int n, uint32_t (*wave_cnt)[256],
                                          uint32_t* wsum, int* flag) {
  uint2* cur = a;
  uint2* oth = b;
#pragma unroll 1
  for (int shift = 0; shift < 32; shift += 8) {
    if (ts_pass<RankT>(cur, oth, ranks, n, shift, wave_cnt, wsum, flag)) {
      uint2* t = cur; cur = oth; oth = t;
    }
    __syncthreads();
  }
  // equal depths must come out in ascending row id (A.3: stable sort of an id-ordered emission): odd-even
  // transposition restricted to equal keys, until a full round moves nothing (one round when there are no ties)
  for (;;) {
    int moved = 0;
#pragma unroll 1
    for (int phase = 0; phase < 2; ++phase) {
      for (int i = 2 * threadIdx.x + phase; i + 1 < n; i += 2 * TS_THREADS) {
        const uint2 lo = cur[i], hi = cur[i + 1];
        if (lo.x == hi.x && lo.y > hi.y) {
          cur[i] = hi;
          cur[i + 1] = lo;
          moved = 1;
        }
      }
      __syncthreads();
    }
    if (!__syncthreads_or(moved)) break;
  }
  return cur;
}

__global__ __launch_bounds__(TS_THREADS) void tile_sort_kernel(int32_t T, int32_t tile_n_bits, int64_t n_tiles,
                                                               const int32_t* __restrict__ offsets, int64_t n_isects,
                                                               uint2* pairs, int64_t* isect_ids, int32_t* flatten_ids) {
  __shared__ uint2 buf_a[TS_CAP];
  __shared__ uint2 buf_b[TS_CAP];
  __shared__ uint16_t rank_s[TS_CAP];
  __shared__ uint32_t wave_cnt[TS_WAVES][256];
  __shared__ uint32_t wsum[TS_WAVES];
  __shared__ int flag;
  const int64_t tile = blockIdx.x;
  const int start = offsets[tile];
  const int end = (tile + 1 < n_tiles) ? offsets[tile + 1] : (int)n_isects;
  const int n = end - start;
  if (n <= 0) return;
  const int64_t cam = tile / T, tl = tile - cam * T;
  const int64_t key_hi = (cam << (32 + tile_n_bits)) | (tl << 32);
  if (n <= TS_CAP) {
    for (int i = threadIdx.x; i < n; i += TS_THREADS) buf_a[i] = pairs[start + i];
    __syncthreads();
    const uint2* cur = ts_sort<uint16_t>(buf_a, buf_b, rank_s, n, wave_cnt, wsum, &flag);
    for (int i = threadIdx.x; i < n; i += TS_THREADS) {
      const uint2 e = cur[i];
      isect_ids[start + i] = key_hi | (int64_t)e.x;
      flatten_ids[start + i] = (int32_t)e.y;
    }
  } else {
    // long segment: same algorithm on global scratch -- the pair array, the (not yet written) isect_ids segment as the
    // second pair buffer, the flatten_ids segment as the rank array.  One workgroup owns the whole segment.
    uint2* ga = pairs + start;
    uint2* gb = reinterpret_cast<uint2*>(isect_ids + start);
    uint32_t* gr = reinterpret_cast<uint32_t*>(flatten_ids + start);
    const uint2* cur = ts_sort<uint32_t>(ga, gb, gr, n, wave_cnt, wsum, &flag);
    // every thread converts its own entries; entry i of either buffer maps onto output slot i
    for (int i = threadIdx.x; i < n; i += TS_THREADS) {
      const uint2 e = cur[i];
      isect_ids[start + i] = key_hi | (int64_t)e.x;
      flatten_ids[start + i] = (int32_t)e.y;
    }
  }
}

static inline int bin_ilog2_floor(int64_t v) {
  int b = 0;
  while (v >>= 1) ++b;
  return b;
}

}  // namespace gs

static int bin_check(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, int32_t n_slices) {
  if (C < 1 || N < 0 || tile_w < 1 || tile_h < 1 || n_slices < 1) return GS_ERR_INVALID_ARG;
  if ((int64_t)tile_w * tile_h > gs::BIN_MAX_TILES) return GS_ERR_UNSUPPORTED;
  if ((int64_t)C * N >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (C > 65535 || n_slices > 65535) return GS_ERR_UNSUPPORTED;
  return GS_OK;
}

extern "C" int32_t gs_isect_bin_max_tiles(void) { return gs::BIN_MAX_TILES; }

extern "C" int gs_isect_bin_count(int32_t C, int32_t N, int32_t tile_w, int32_t tile_h, const float* splats,
                                  const int32_t* tiles_per_gauss, int32_t n_slices, uint32_t* table, void* stream) {
  int st = bin_check(C, N, tile_w, tile_h, n_slices);
  if (st != GS_OK) return st;
  if (!table || (N > 0 && (!splats || !tiles_per_gauss))) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  int rows = (int)gs::ceil_div(N > 0 ? N : 1, n_slices);
  rows = (int)gs::ceil_div(rows, gs::BIN_THREADS) * gs::BIN_THREADS;
  hipLaunchKernelGGL((gs::isect_bin_kernel<false>), dim3((unsigned)n_slices, (unsigned)C), dim3(gs::BIN_THREADS),
                     (size_t)T * sizeof(uint32_t), (hipStream_t)stream, N, tile_w, T, n_slices, rows, splats,
                     tiles_per_gauss, table, (const int32_t*)nullptr, (uint2*)nullptr);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_bin_offsets(int32_t C, int32_t tile_w, int32_t tile_h, int32_t n_slices, uint32_t* table,
                                    int32_t* tile_counts, int32_t* isect_offsets, int64_t* total, void* stream) {
  if (C < 1 || tile_w < 1 || tile_h < 1 || n_slices < 1) return GS_ERR_INVALID_ARG;
  if (!table || !tile_counts || !isect_offsets || !total) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  const int64_t n_tiles = (int64_t)C * T;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs::bin_slice_prefix_kernel, dim3((unsigned)gs::ceil_div(n_tiles, 256)), dim3(256), 0, s, C, T,
                     n_slices, table, tile_counts);
  hipLaunchKernelGGL(gs::bin_offsets_kernel, dim3(1), dim3(1024), 0, s, n_tiles, tile_counts, isect_offsets, total);
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
  int rows = (int)gs::ceil_div(N > 0 ? N : 1, n_slices);
  rows = (int)gs::ceil_div(rows, gs::BIN_THREADS) * gs::BIN_THREADS;
  hipLaunchKernelGGL((gs::isect_bin_kernel<true>), dim3((unsigned)n_slices, (unsigned)C), dim3(gs::BIN_THREADS),
                     (size_t)T * sizeof(uint32_t), (hipStream_t)stream, N, tile_w, T, n_slices, rows, splats,
                     tiles_per_gauss, const_cast<uint32_t*>(table), isect_offsets, reinterpret_cast<uint2*>(pairs));
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_isect_tile_sort(int32_t C, int32_t tile_w, int32_t tile_h, const int32_t* isect_offsets,
                                  int64_t n_isects, uint64_t* pairs, int64_t* isect_ids, int32_t* flatten_ids,
                                  void* stream) {
  if (C < 1 || tile_w < 1 || tile_h < 1 || n_isects < 0) return GS_ERR_INVALID_ARG;
  if (n_isects == 0) return GS_OK;
  if (n_isects >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (!isect_offsets || !pairs || !isect_ids || !flatten_ids) return GS_ERR_INVALID_ARG;
  const int T = tile_w * tile_h;
  const int tile_n_bits = gs::bin_ilog2_floor(T) + 1;
  const int cam_n_bits = gs::bin_ilog2_floor(C) + 1;
  if (32 + tile_n_bits + cam_n_bits > 64) return GS_ERR_UNSUPPORTED;
  const int64_t n_tiles = (int64_t)C * T;
  hipLaunchKernelGGL(gs::tile_sort_kernel, dim3((unsigned)n_tiles), dim3(gs::TS_THREADS), 0, (hipStream_t)stream, T,
                     tile_n_bits, n_tiles, isect_offsets, n_isects, reinterpret_cast<uint2*>(pairs), isect_ids,
                     flatten_ids);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
