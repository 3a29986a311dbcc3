// Stable LSD radix sort of (uint64 key, uint32 value) pairs on the key bits [begin_bit, end_bit).
// Replaces the cub::DeviceRadixSort::SortPairs<int64,int32> call inside gsplat's isect_tiles
// (SURVEY.md section 8(a) row 6, Appendix A.3): ties keep their input order.
//
// gfx950 design.  Digits are up to 9 bits wide and the bit range is split evenly over the fewest passes
// (35 key bits of the depth pre-sort -> 4 passes, 14 (camera|tile) bits -> 2 passes of 7).  Three launches per pass:
//   (1) rs_hist     per-workgroup digit histogram into a digit-major table [radix][blocks]
//   (2) rs_colscan  one wave per digit: exclusive scan of its table row over the blocks (+ the digit total)
//   (3) rs_scatter  exclusive scan of the digit totals in LDS, stable ranking, scatter
// In-workgroup ranking is wave64 match-any: nine 64-bit ballots give each lane the mask of lanes that hold the
// same digit, so a rank is one popcount plus one per-wave LDS counter -- no per-thread counter arrays and no
// LDS-atomic pile-up when a whole wave holds one digit (the (camera|tile) passes do).
// The element count may live in device memory (`n_dev`): grids are sized for `n_max` and surplus workgroups exit,
// which lets the host enqueue a sort whose size it does not know yet.  HBM-bound integer work.
#include <mutex>

#include "gs_common.hpp"

namespace gs {

constexpr int RS_THREADS = 256;
constexpr int RS_WAVES = RS_THREADS / 64;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup
constexpr int RS_MAX_BITS = 9;
constexpr int RS_MAX_RADIX = 1 << RS_MAX_BITS;

// lanes (among the valid ones) whose digit equals this lane's digit
__device__ __forceinline__ unsigned long long rs_peers(uint32_t d, bool valid) {
  unsigned long long peers = __ballot(valid);
#pragma unroll
  for (int b = 0; b < RS_MAX_BITS; ++b) {
    const bool bit = (d >> b) & 1u;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

__device__ __forceinline__ int64_t rs_count(const int64_t* __restrict__ n_dev, int64_t n_max) {
  if (!n_dev) return n_max;
  const int64_t n = *n_dev;
  return n < n_max ? n : n_max;
}

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t* __restrict__ keys,
                                                             const int64_t* __restrict__ n_dev, int64_t n_max,
                                                             int shift, uint32_t digit_mask, int64_t nb_stride,
                                                             uint32_t* __restrict__ table) {
  const int64_t n = rs_count(n_dev, n_max);
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  if (base >= n) return;
  __shared__ uint32_t hist[RS_MAX_RADIX];
  hist[threadIdx.x] = 0;
  hist[threadIdx.x + RS_THREADS] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t wbase = base + (int64_t)wave * (64 * RS_ITEMS);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint64_t key[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = wbase + i * 64 + lane;
    key[i] = (pos < n) ? keys[pos] : 0ull;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const bool valid = wbase + i * 64 + lane < n;
    const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
    const unsigned long long peers = rs_peers(d, valid);
    if (valid && (peers & lt_mask) == 0ull) atomicAdd(&hist[d], (uint32_t)__popcll(peers));
  }
  __syncthreads();
  const uint32_t radix = digit_mask + 1u;
  for (uint32_t d = threadIdx.x; d < radix; d += RS_THREADS) table[(int64_t)d * nb_stride + blockIdx.x] = hist[d];
}

// one wave per digit: exclusive scan over the active blocks of the digit's table row, total -> totals[d]
__global__ __launch_bounds__(RS_THREADS) void rs_colscan_kernel(const int64_t* __restrict__ n_dev, int64_t n_max,
                                                                uint32_t radix, int64_t nb_stride,
                                                                uint32_t* __restrict__ table,
                                                                uint32_t* __restrict__ totals) {
  const int lane = threadIdx.x & 63;
  const uint32_t d = blockIdx.x * RS_WAVES + (threadIdx.x >> 6);
  if (d >= radix) return;
  const int64_t n = rs_count(n_dev, n_max);
  const int64_t nb = ceil_div(n, RS_TILE);
  uint32_t* row = table + (int64_t)d * nb_stride;
  uint32_t run = 0;
  for (int64_t c0 = 0; c0 < nb; c0 += 64) {
    const int64_t c = c0 + lane;
    const uint32_t v = (c < nb) ? row[c] : 0u;
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (c < nb) row[c] = run + incl - v;
    run += __shfl(incl, 63);
  }
  if (lane == 0) totals[d] = run;
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint64_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, const int64_t* __restrict__ n_dev, int64_t n_max, int shift, uint32_t digit_mask,
    int64_t nb_stride, const uint32_t* __restrict__ table, const uint32_t* __restrict__ totals) {
  const int64_t n = rs_count(n_dev, n_max);
  const int64_t base = (int64_t)blockIdx.x * RS_TILE;
  if (base >= n) return;
  __shared__ uint32_t wave_cnt[RS_WAVES][RS_MAX_RADIX];
  __shared__ uint32_t dbase[RS_MAX_RADIX];
  __shared__ uint32_t wsum[RS_WAVES];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t radix = digit_mask + 1u;
#pragma unroll
  for (int w = 0; w < RS_WAVES; ++w) {
    wave_cnt[w][threadIdx.x] = 0;
    wave_cnt[w][threadIdx.x + RS_THREADS] = 0;
  }
  // exclusive scan of the digit totals: thread t owns digits 2t, 2t+1
  const uint32_t t0 = (2 * threadIdx.x < radix) ? totals[2 * threadIdx.x] : 0u;
  const uint32_t t1 = (2 * threadIdx.x + 1 < radix) ? totals[2 * threadIdx.x + 1] : 0u;
  {
    uint32_t incl = t0 + t1;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w)
      if (w < wave) before += wsum[w];
    const uint32_t ex = before + incl - (t0 + t1);
    dbase[2 * threadIdx.x] = ex;
    dbase[2 * threadIdx.x + 1] = ex + t0;
  }

  const int64_t wbase = base + (int64_t)wave * (64 * RS_ITEMS);
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint64_t key[RS_ITEMS];
  uint32_t val[RS_ITEMS];
  uint32_t rank[RS_ITEMS];
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const int64_t pos = wbase + i * 64 + lane;
    key[i] = (pos < n) ? keys_in[pos] : 0ull;
    val[i] = (pos < n) ? vals_in[pos] : 0u;
  }
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    const bool valid = wbase + i * 64 + lane < n;
    const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
    const unsigned long long peers = rs_peers(d, valid);
    const uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t prev = wave_cnt[wave][d];
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0) wave_cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    __builtin_amdgcn_wave_barrier();
    rank[i] = prev + before;
  }
  __syncthreads();
  // per-wave counts -> per-wave global start offsets
  for (uint32_t d = threadIdx.x; d < radix; d += RS_THREADS) {
    uint32_t run = dbase[d] + table[(int64_t)d * nb_stride + blockIdx.x];
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) {
      const uint32_t c = wave_cnt[w][d];
      wave_cnt[w][d] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < RS_ITEMS; ++i) {
    if (wbase + i * 64 + lane < n) {
      const uint32_t d = (uint32_t)(key[i] >> shift) & digit_mask;
      const uint32_t dst = wave_cnt[wave][d] + rank[i];
      keys_out[dst] = key[i];
      vals_out[dst] = val[i];
    }
  }
}

}  // namespace gs

static int64_t rs_table_bytes(int64_t n);

extern "C" int64_t gs_radix_sort_workspace_bytes(int64_t n) {
  if (n <= 0) return 256;
  return rs_table_bytes(n) + gs::RS_MAX_RADIX * (int64_t)sizeof(uint32_t);
}

// The passes of one sort.  `table` holds nb * RS_MAX_RADIX counters (digit-major, row stride nb), `totals` RS_MAX_RADIX.
static int rs_run(int64_t n_max, const int64_t* n_dev, int32_t begin_bit, int32_t end_bit, uint64_t* keys_a,
                  uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, uint32_t* table, uint32_t* totals,
                  int32_t* result_in_b_host, hipStream_t s) {
  const int64_t nb = gs::ceil_div(n_max, gs::RS_TILE);
  const int total_bits = end_bit - begin_bit;
  const int npass = (total_bits + gs::RS_MAX_BITS - 1) / gs::RS_MAX_BITS;
  const int bits_per_pass = (total_bits + npass - 1) / npass;

  uint64_t* kin = keys_a; uint32_t* vin = vals_a;
  uint64_t* kout = keys_b; uint32_t* vout = vals_b;
  int in_b = 0;
  for (int shift = begin_bit; shift < end_bit; shift += bits_per_pass) {
    const int nbits = (end_bit - shift < bits_per_pass) ? (end_bit - shift) : bits_per_pass;
    const uint32_t mask = (1u << nbits) - 1u;
    const uint32_t radix = mask + 1u;
    hipLaunchKernelGGL(gs::rs_hist_kernel, dim3((unsigned)nb), dim3(gs::RS_THREADS), 0, s, kin, n_dev, n_max, shift,
                       mask, nb, table);
    hipLaunchKernelGGL(gs::rs_colscan_kernel, dim3((radix + gs::RS_WAVES - 1) / gs::RS_WAVES), dim3(gs::RS_THREADS), 0,
                       s, n_dev, n_max, radix, nb, table, totals);
    hipLaunchKernelGGL(gs::rs_scatter_kernel, dim3((unsigned)nb), dim3(gs::RS_THREADS), 0, s, kin, vin, kout, vout,
                       n_dev, n_max, shift, mask, nb, table, totals);
    GS_CHECK_LAUNCH();
    uint64_t* tk = kin; kin = kout; kout = tk;
    uint32_t* tv = vin; vin = vout; vout = tv;
    in_b ^= 1;
  }
  *result_in_b_host = in_b;
  return GS_OK;
}

static int64_t rs_table_bytes(int64_t n) {
  const int64_t nb = gs::ceil_div(n, gs::RS_TILE);
  return ((nb * gs::RS_MAX_RADIX * (int64_t)sizeof(uint32_t) + 255) / 256) * 256;
}

extern "C" int gs_radix_sort_pairs_u64_u32_ex(int64_t n_max, const int64_t* n_dev, int32_t begin_bit, int32_t end_bit,
                                              uint64_t* keys_a, uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b,
                                              void* workspace, int64_t workspace_bytes, int32_t* result_in_b_host,
                                              void* stream) {
  if (n_max < 0 || begin_bit < 0 || end_bit < begin_bit || end_bit > 64 || !result_in_b_host) return GS_ERR_INVALID_ARG;
  *result_in_b_host = 0;
  if (n_max == 0 || end_bit == begin_bit) return GS_OK;
  if (n_max >= ((int64_t)1 << 31)) return GS_ERR_UNSUPPORTED;
  if (!keys_a || !vals_a || !keys_b || !vals_b || !workspace) return GS_ERR_INVALID_ARG;
  if (workspace_bytes < gs_radix_sort_workspace_bytes(n_max)) return GS_ERR_WORKSPACE;
  uint32_t* table = (uint32_t*)workspace;
  uint32_t* totals = (uint32_t*)((char*)workspace + rs_table_bytes(n_max));
  return rs_run(n_max, n_dev, begin_bit, end_bit, keys_a, vals_a, keys_b, vals_b, table, totals, result_in_b_host,
                (hipStream_t)stream);
}

// ---- the same sort for callers that cannot allocate (include/gsplat_mi355x.h, gs_radix_sort_pairs_u64_u32_static) ----
// The digit table and the digit totals live in a file-scope device array instead of a caller workspace.  Every GPU the
// library runs on holds its own copy of the array (8 MiB + 2 KiB until the process ends) at its own address; the host
// gets that address once per device from hipGetSymbolAddress and passes it to the kernels like any other workspace --
// the array's NAME is a host-side stand-in and is never handed to a kernel.  Nothing in the table has to start at
// zero: every pass's histogram kernel overwrites each counter the scan and the scatter of that pass read.
namespace {
constexpr int64_t RS_STATIC_MAX_N = GS_RADIX_SORT_STATIC_MAX_N;
constexpr int64_t RS_STATIC_TABLE_BYTES = (RS_STATIC_MAX_N / gs::RS_TILE) * gs::RS_MAX_RADIX * (int64_t)sizeof(uint32_t);
constexpr int64_t RS_STATIC_BYTES = RS_STATIC_TABLE_BYTES + gs::RS_MAX_RADIX * (int64_t)sizeof(uint32_t);
static_assert(RS_STATIC_MAX_N % gs::RS_TILE == 0 && RS_STATIC_TABLE_BYTES % 256 == 0, "table size");
constexpr int RS_MAX_DEVICES = 64;
}  // namespace

static __device__ __align__(256) unsigned char scratch[RS_STATIC_BYTES];

static void* g_scratch_ptr[RS_MAX_DEVICES];  // per device: address of that device's copy of `scratch`
static std::mutex g_scratch_mutex;

// Address of the current device's `scratch` (looked up once per device, with that device current).  The stream's
// device must be the current one: the launches go to the stream's device, and another device's copy of the array
// lives at another address (a peer write, or a fault).
static int rs_static_scratch(void** out, hipStream_t stream) {
  int dev = -1;
  GS_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= RS_MAX_DEVICES) return GS_ERR_UNSUPPORTED;
  if (stream) {
    int sdev = -1;
    GS_HIP(hipStreamGetDevice(stream, &sdev));
    if (sdev != dev) return GS_ERR_INVALID_ARG;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    void* ptr = nullptr;
    GS_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return GS_ERR_WORKSPACE;
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return GS_OK;
}

extern "C" int gs_radix_sort_pairs_u64_u32_static(int64_t n, int32_t begin_bit, int32_t end_bit, uint64_t* keys_a,
                                                  uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b,
                                                  int32_t* result_in_b_host, void* stream) {
  if (n < 0 || begin_bit < 0 || end_bit < begin_bit || end_bit > 64 || !result_in_b_host) return GS_ERR_INVALID_ARG;
  *result_in_b_host = 0;
  if (n > RS_STATIC_MAX_N) return GS_ERR_WORKSPACE;  // beyond the static table: use the workspace entry points
  // (before the empty-input return: a call with n == 0 is the warm-up that resolves this device's scratch address --
  // the first lookup may load the code object and is not for a stream under capture)
  void* ptr = nullptr;
  const int st = rs_static_scratch(&ptr, (hipStream_t)stream);
  if (st != GS_OK) return st;
  if (n == 0 || end_bit == begin_bit) return GS_OK;
  if (!keys_a || !vals_a || !keys_b || !vals_b) return GS_ERR_INVALID_ARG;
  uint32_t* table = (uint32_t*)ptr;
  uint32_t* totals = (uint32_t*)((char*)ptr + RS_STATIC_TABLE_BYTES);
  return rs_run(n, nullptr, begin_bit, end_bit, keys_a, vals_a, keys_b, vals_b, table, totals, result_in_b_host,
                (hipStream_t)stream);
}

extern "C" int gs_radix_sort_pairs_u64_u32_bits(int64_t n, int32_t begin_bit, int32_t end_bit, uint64_t* keys_a,
                                                uint32_t* vals_a, uint64_t* keys_b, uint32_t* vals_b, void* workspace,
                                                int64_t workspace_bytes, int32_t* result_in_b_host, void* stream) {
  return gs_radix_sort_pairs_u64_u32_ex(n, nullptr, begin_bit, end_bit, keys_a, vals_a, keys_b, vals_b, workspace,
                                        workspace_bytes, result_in_b_host, stream);
}

extern "C" int gs_radix_sort_pairs_u64_u32(int64_t n, int32_t end_bit, uint64_t* keys_a, uint32_t* vals_a,
                                           uint64_t* keys_b, uint32_t* vals_b, void* workspace,
                                           int64_t workspace_bytes, int32_t* result_in_b_host, void* stream) {
  return gs_radix_sort_pairs_u64_u32_ex(n, nullptr, 0, end_bit, keys_a, vals_a, keys_b, vals_b, workspace,
                                        workspace_bytes, result_in_b_host, stream);
}
