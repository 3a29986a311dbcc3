// Stable LSD radix sort of (uint64 key, uint32 value) pairs on the low `end_bit` key bits.
// Replaces the cub::DeviceRadixSort::SortPairs<int64,int32> call inside gsplat's isect_tiles
// (SURVEY.md section 8(a) row 6, Appendix A.3): ties keep emission order, i.e. ascending flatten id.
//
// gfx950 design: 8-bit digits; per pass  (1) per-workgroup digit histogram (LDS atomics),
// (2) exclusive scan of the digit-major [256][nblocks] table, (3) stable scatter.  The in-workgroup
// ranking is wave64 match-any: eight 64-bit ballots give each lane the mask of lanes holding the same
// digit, so a rank is one popcount and one per-wave LDS counter read -- no per-thread counter arrays.
// Only the bits actually used by the (camera|tile|depth) key are sorted.  HBM-bound integer work.
#include "gs_common.hpp"

namespace gs {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_BITS = 8;
constexpr int RS_RADIX = 1 << RS_BITS;

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int shift,
                                                             uint32_t digit_mask, int64_t nblocks,
                                                             int32_t* __restrict__ block_hist) {
  __shared__ uint32_t hist[RS_RADIX];
  hist[threadIdx.x] = 0;
  __syncthreads();
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = base + (int64_t)i * RS_THREADS + threadIdx.x;
    if (pos < n) {
      const uint32_t d = (uint32_t)(keys[pos] >> shift) & digit_mask;
      atomicAdd(&hist[d], 1u);
    }
  }
  __syncthreads();
  block_hist[(int64_t)threadIdx.x * nblocks + blockIdx.x] = (int32_t)hist[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, int64_t n, int shift, uint32_t digit_mask, int64_t nblocks,
    const int32_t* __restrict__ block_offsets) {
  __shared__ uint32_t wave_cnt[RS_WAVES][RS_RADIX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) wave_cnt[w][threadIdx.x] = 0;
  __syncthreads();

  const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (64 * RS_ITEMS);
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  uint64_t key[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = wbase + i * 64 + lane;
    key[i] = (pos < n) ? keys_in[pos] : ~0ull;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = wbase + i * 64 + lane;
    const bool valid = pos < n;
    const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < RS_BITS; ++b) {
      const bool bit = (d >> b) & 1u;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t prev = wave_cnt[wave][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wave_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[i] = prev + before;
  }
  __syncthreads();
  {
    // digit d = threadIdx.x: turn per-wave counts into per-wave global start offsets
    uint32_t run = (uint32_t)block_offsets[(int64_t)threadIdx.x * nblocks + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const uint32_t c = wave_cnt[w][threadIdx.x];
      wave_cnt[w][threadIdx.x] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = wbase + i * 64 + lane;
    if (pos < n) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
      const uint32_t dst = wave_cnt[wave][d] + rank[i];
      keys_out[dst] = key[i];
      vals_out[dst] = vals_in[pos];
    }
  }
}

}  // namespace gs

extern "C" int64_t gs_radix_sort_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  const int64_t nb = gs::ceil_div(n, gs::RS_TILE);
  const int64_t table = ((nb * gs::RS_RADIX * (int64_t)sizeof(int32_t) + 255) / 256) * 256;
  return 2 * table + 256 + gs_exclusive_scan_workspace_bytes(nb * gs::RS_RADIX);
}

extern "C" int gs_radix_sort_pairs_u64_u32(int64_t n, int32_t end_bit, uint64_t* keys_a, uint32_t* vals_a,
                                           uint64_t* keys_b, uint32_t* vals_b, void* workspace,
                                           int64_t workspace_bytes, int32_t* result_in_b_host, void* stream) {
  if (n < 0 || end_bit < 0 || end_bit > 64 || !result_in_b_host) return GS_ERR_INVALID_ARG;
  *result_in_b_host = 0;
  if (n == 0 || end_bit == 0) return GS_OK;
  if (n >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (!keys_a || !vals_a || !keys_b || !vals_b || !workspace) return GS_ERR_INVALID_ARG;
  if (workspace_bytes < gs_radix_sort_workspace_bytes(n)) return GS_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = gs::ceil_div(n, gs::RS_TILE);
  const int64_t table = ((nb * gs::RS_RADIX * (int64_t)sizeof(int32_t) + 255) / 256) * 256;
  char* ws = (char*)workspace;
  int32_t* hist = (int32_t*)ws;
  int32_t* offs = (int32_t*)(ws + table);
  int64_t* total = (int64_t*)(ws + 2 * table);
  void* scan_ws = ws + 2 * table + 256;
  const int64_t scan_ws_bytes = workspace_bytes - (2 * table + 256);

  uint64_t* kin = keys_a; uint32_t* vin = vals_a;
  uint64_t* kout = keys_b; uint32_t* vout = vals_b;
  int in_b = 0;
  for (int shift = 0; shift < end_bit; shift += gs::RS_BITS) {
    const int nbits = (end_bit - shift < gs::RS_BITS) ? (end_bit - shift) : gs::RS_BITS;
    const uint32_t mask = (1u << nbits) - 1u;
    hipLaunchKernelGGL(gs::rs_hist_kernel, dim3((unsigned)nb), dim3(gs::RS_THREADS), 0, s, kin, n, shift, mask, nb, hist);
    int st = gs_exclusive_scan_i32(hist, offs, nb * gs::RS_RADIX, total, scan_ws, scan_ws_bytes, stream);
    if (st != GS_OK) return st;
    hipLaunchKernelGGL(gs::rs_scatter_kernel, dim3((unsigned)nb), dim3(gs::RS_THREADS), 0, s, kin, vin, kout, vout, n,
                       shift, mask, nb, offs);
    GS_CHECK_LAUNCH();
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
    in_b ^= 1;
  }
  *result_in_b_host = in_b;
  return GS_OK;
}
