// Standalone spherical_harmonics forward/backward (HBM-bound: 12+12K B read, 12 B written per row; the reference calls
// it once per Gaussian class and step: models/gaussians/vanilla.py:369, models/nodes/rigid.py:486, ...).
// One thread per (flattened) Gaussian row.  The backward's coefficient-gradient rows are dense (masked-out rows are
// zeros) and leave wave-cooperatively through LDS as contiguous 256-B wave rows (sh_math.hpp, wave_rows_store):
// 3.6 -> 6.4 TB/s at 3 M rows, K = 16.
#include "sh_math.hpp"

namespace gs {

template <int DEG>
struct ShTile {
  static constexpr int NB3 = (DEG + 1) * (DEG + 1) * 3;
  static constexpr int ROWS = (NB3 <= 48) ? 64 : 32;  // rows per round: 4 waves x ROWS x (NB3 + 1) floats of LDS
  static constexpr int FLOATS = ROWS * (NB3 + 1);
};

template <int DEG>
__global__ __launch_bounds__(256) void sh_fwd_kernel(int64_t n, int K, const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const uint8_t* __restrict__ masks,
                                                     float* __restrict__ out) {
  // each lane streams its own coefficient row as aligned float4s, all issued back to back (sh_load_row: the compiler's own
  // merge of `coeffs[3 k + c]` is a 4-B-aligned dwordx4 / dwordx3 mix that streams a third slower); the 12-B dirs / out
  // rows move as the wave's 768 contiguous bytes (gs_common.hpp).  With fewer bands in use than K a lane only touches
  // the lines it needs.
  __shared__ float s_tile[4][192];
  constexpr int NB3 = 3 * (DEG + 1) * (DEG + 1);
  const int64_t i_raw = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = i_raw - lane;
  const int rows = (int)min((int64_t)64, n - row0);  // wave-uniform
  if (rows <= 0) return;
  const bool live = i_raw < n;
  const int64_t i = live ? i_raw : n - 1;
  float* tile = s_tile[threadIdx.x >> 6];
  const bool on = masks == nullptr || masks[i];
  float creg[NB3];
#pragma unroll
  for (int k = 0; k < NB3; ++k) creg[k] = 0.f;
  if (on) sh_load_row<NB3>(coeffs + i * K * 3, (K % 4 == 0) && aligned16(coeffs), creg);
  float d[3];
  if (aligned16(dirs)) {
    wave_load3(dirs, row0, rows, lane, tile, d);
  } else {
    d[0] = dirs[3 * i]; d[1] = dirs[3 * i + 1]; d[2] = dirs[3 * i + 2];
  }
  float o[3] = {0.f, 0.f, 0.f};
  if (on) sh_eval<DEG>(creg, d[0], d[1], d[2], o);
  if (aligned16(out)) {
    wave_store3(out, row0, rows, lane, tile, o);
  } else if (live) {
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
  }
}

template <int DEG, bool DIRGRAD>
__global__ __launch_bounds__(256) void sh_bwd_kernel(int64_t n, int K, const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const uint8_t* __restrict__ masks,
                                                     const float* __restrict__ v_out,
                                                     float* __restrict__ v_coeffs,
                                                     float* __restrict__ v_dirs) {
  using TL = ShTile<DEG>;
  constexpr int NB = (DEG + 1) * (DEG + 1);
  __shared__ float s_t[4 * TL::FLOATS];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = i < n;
  float acc[NB * 3];
#pragma unroll
  for (int k = 0; k < NB * 3; ++k) acc[k] = 0.f;
  float vx = 0.f, vy = 0.f, vz = 0.f;
  if (live && (masks == nullptr || masks[i])) {
    const float v[3] = {v_out[3 * i], v_out[3 * i + 1], v_out[3 * i + 2]};
    sh_eval_vjp<DEG, DIRGRAD>(coeffs + i * K * 3, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], v, acc, vx,
                              vy, vz);
  }
  // v_coeffs is dense (masked-out rows are zeros): always the cooperative store
  float* t = s_t + wave * TL::FLOATS;
  const int64_t row_w = i - lane;
#pragma unroll 1
  for (int h = 0; h < 64 / TL::ROWS; ++h) {
    if (lane / TL::ROWS == h) {
#pragma unroll
      for (int k = 0; k < NB * 3; ++k) t[(lane % TL::ROWS) * (TL::NB3 + 1) + k] = acc[k];
    }
    const int64_t row0 = row_w + h * TL::ROWS;
    const int rows = (int)min((int64_t)TL::ROWS, n - row0);
    wave_rows_store<TL::NB3>(t, v_coeffs + row0 * K * 3, rows, K * 3, lane);
  }
  if (!live) return;
  if constexpr (DIRGRAD) {
    v_dirs[3 * i] = vx;
    v_dirs[3 * i + 1] = vy;
    v_dirs[3 * i + 2] = vz;
  }
}

// ---- SH backward from packed colour-gradient messages ---------------------------------------------------------------
// The SH half of gs_projection_bwd_split: v_sh [N,K,3] and the view-direction part of v_means from `n_msgs` messages of
// the layout of gs_colour_message_layout (projection.hip), `msg_stride` bytes apart: this rank's own message, or every
// rank's after one all-gather -- the per-(camera, Gaussian) colour gradient is 12 B where the SH-coefficient gradient it
// expands to is 12 K B, which is what makes the exchange of a camera-sharded data-parallel step cheap
// (gsplat/distributed.py).  One thread per Gaussian, looping over messages x cameras in a fixed order, so every rank
// computes bit-identical sums; a camera's visibility word and prefix are wave-uniform, the pair's 12-B value is read at
// prefix + (set bits below the lane): lanes read consecutive rows.
struct MsgLayoutSh {
  int64_t campos, words, prefix, vals;
};

// The Gaussian's coefficient row (needed for the view-direction gradient) is loaded as aligned float4s inside the
// visible branch.  Measured at configs[1] (scratch/run_exp2.sh): 0.097 ms; through the plain pointer 0.096; prefetched
// up front for every Gaussian some camera sees 0.113-0.115 ms -- the prefetch costs a wave per SIMD.
template <int DEG, bool VIEWGRAD>
__global__ __launch_bounds__(256, 1) void sh_bwd_packed_kernel(
    int N, int K, int C, int n_msgs, int64_t msg_stride, int64_t cap, MsgLayoutSh L, const float* __restrict__ means,
    const float* __restrict__ coeffs, const unsigned char* __restrict__ msgs, float* __restrict__ v_coeffs,
    float* __restrict__ v_means, int accumulate_means, int local_msg, float* __restrict__ v_campos) {
  using TL = ShTile<DEG>;
  constexpr int NB = (DEG + 1) * (DEG + 1);
  __shared__ float s_t[4 * TL::FLOATS];
  const int n_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = n_raw < N;
  const int n = live ? n_raw : N - 1;
  const int n_words = (N + 63) >> 6;
  const int w = __builtin_amdgcn_readfirstlane(n_raw >> 6);  // wave-uniform: scalar loads of the words below
  const bool has_word = w < n_words;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const float mu[3] = {means[3 * n], means[3 * n + 1], means[3 * n + 2]};
  float acc[NB * 3];
#pragma unroll
  for (int k = 0; k < NB * 3; ++k) acc[k] = 0.f;
  float vmx = 0.f, vmy = 0.f, vmz = 0.f;
  for (int m = 0; m < n_msgs && has_word; ++m) {
    const unsigned char* msg = msgs + (int64_t)m * msg_stride;
    const uint64_t* words = reinterpret_cast<const uint64_t*>(msg + L.words);
    const int32_t* prefix = reinterpret_cast<const int32_t*>(msg + L.prefix);
    const float* vals = reinterpret_cast<const float*>(msg + L.vals);
    const float* cps = reinterpret_cast<const float*>(msg + L.campos);
    // cameras of THIS message (ranks may hold different numbers of cameras; the layout is sized for the largest)
    const int n_cams = min(max(__builtin_amdgcn_readfirstlane(*reinterpret_cast<const int32_t*>(msg)), 0), C);
    for (int c = 0; c < n_cams; ++c) {
      const unsigned long long word = words[(int64_t)c * n_words + w];
      float gx = 0.f, gy = 0.f, gz = 0.f;
      if (word != 0ull) {  // wave-uniform
        if (live && ((word >> lane) & 1ull)) {
          const int64_t row = (int64_t)prefix[(int64_t)c * n_words + w] + __popcll(word & lt_mask);
          if (row >= 0 && row < cap) {  // (always, for a well-formed message: cap >= its visible pairs)
            const float v[3] = {vals[3 * row], vals[3 * row + 1], vals[3 * row + 2]};
            float creg[NB * 3];
            sh_load_row<NB * 3>(coeffs + (int64_t)n * K * 3, (K % 4 == 0) && aligned16(coeffs), creg);
            sh_eval_vjp<DEG, true>(creg, mu[0] - cps[3 * c], mu[1] - cps[3 * c + 1], mu[2] - cps[3 * c + 2], v, acc,
                                   gx, gy, gz);
            vmx += gx; vmy += gy; vmz += gz;
          }
        }
        if constexpr (VIEWGRAD) {
          // d colour / d camera centre = -d colour / d mean, summed over the Gaussians (this rank's cameras only)
          if (m == local_msg) {
            float sx = -gx, sy = -gy, sz = -gz;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
              sx += __shfl_xor(sx, o); sy += __shfl_xor(sy, o); sz += __shfl_xor(sz, o);
            }
            if (lane == 0) {
              atomicAdd(&v_campos[3 * c + 0], sx);
              atomicAdd(&v_campos[3 * c + 1], sy);
              atomicAdd(&v_campos[3 * c + 2], sz);
            }
          }
        }
      }
    }
  }
  float* t = s_t + wave * TL::FLOATS;
  const int row_w = n_raw - lane;
#pragma unroll 1
  for (int h = 0; h < 64 / TL::ROWS; ++h) {
    if (lane / TL::ROWS == h) {
#pragma unroll
      for (int k = 0; k < NB * 3; ++k) t[(lane % TL::ROWS) * (TL::NB3 + 1) + k] = acc[k];
    }
    const int row0 = row_w + h * TL::ROWS;
    const int rows = min(TL::ROWS, N - row0);
    wave_rows_store<TL::NB3>(t, v_coeffs + (int64_t)row0 * K * 3, rows, K * 3, lane);
  }
  if (!live) return;
  if (accumulate_means) {
    v_means[3 * n] += vmx; v_means[3 * n + 1] += vmy; v_means[3 * n + 2] += vmz;
  } else {
    v_means[3 * n] = vmx; v_means[3 * n + 1] = vmy; v_means[3 * n + 2] = vmz;
  }
}

}  // namespace gs

extern "C" int gs_colour_message_layout(int32_t C, int32_t N, int64_t cap, int64_t* out5_host);

extern "C" int gs_sh_bwd_packed(int32_t N, int32_t K, int32_t degree, int32_t C, int32_t n_msgs, int64_t msg_stride,
                                int64_t cap, const float* means, const float* coeffs, const void* messages,
                                float* v_coeffs, float* v_means, int32_t accumulate_means, int32_t local_msg,
                                float* v_campos, void* stream) {
  if (N < 0 || degree < 0 || degree > 4 || (degree + 1) * (degree + 1) > K || C < 1 || n_msgs < 1 || cap < 0)
    return GS_ERR_INVALID_ARG;
  if (N == 0) return GS_OK;
  int64_t lay[5];
  int st = gs_colour_message_layout(C, N, cap, lay);
  if (st != GS_OK) return st;
  if (!means || !coeffs || !messages || !v_coeffs || !v_means || msg_stride < lay[4]) return GS_ERR_INVALID_ARG;
  if (v_campos && (local_msg < 0 || local_msg >= n_msgs)) return GS_ERR_INVALID_ARG;
  const gs::MsgLayoutSh L{lay[0], lay[1], lay[2], lay[3]};
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(N, 256)), block(256);
  const unsigned char* msgs = (const unsigned char*)messages;
#define GS_SHP_CASE(D)                                                                                              \
  case D:                                                                                                           \
    if (v_campos)                                                                                                   \
      hipLaunchKernelGGL((gs::sh_bwd_packed_kernel<D, true>), grid, block, 0, s, N, K, C, n_msgs, msg_stride, cap, L, \
                         means, coeffs, msgs, v_coeffs, v_means, accumulate_means, local_msg, v_campos);            \
    else                                                                                                            \
      hipLaunchKernelGGL((gs::sh_bwd_packed_kernel<D, false>), grid, block, 0, s, N, K, C, n_msgs, msg_stride, cap, L, \
                         means, coeffs, msgs, v_coeffs, v_means, accumulate_means, local_msg, v_campos);            \
    break;
  switch (degree) {
    GS_SHP_CASE(0)
    GS_SHP_CASE(1)
    GS_SHP_CASE(2)
    GS_SHP_CASE(3)
    default:
      GS_SHP_CASE(4)
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_sh_fwd(int64_t n, int32_t K, int32_t degree, const float* dirs, const float* coeffs,
                         const uint8_t* masks, float* out, void* stream) {
  if (n < 0 || degree < 0 || degree > 4 || (degree + 1) * (degree + 1) > K) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!dirs || !coeffs || !out) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(n, 256)), block(256);
  switch (degree) {
    case 0: hipLaunchKernelGGL(gs::sh_fwd_kernel<0>, grid, block, 0, s, n, K, dirs, coeffs, masks, out); break;
    case 1: hipLaunchKernelGGL(gs::sh_fwd_kernel<1>, grid, block, 0, s, n, K, dirs, coeffs, masks, out); break;
    case 2: hipLaunchKernelGGL(gs::sh_fwd_kernel<2>, grid, block, 0, s, n, K, dirs, coeffs, masks, out); break;
    case 3: hipLaunchKernelGGL(gs::sh_fwd_kernel<3>, grid, block, 0, s, n, K, dirs, coeffs, masks, out); break;
    default: hipLaunchKernelGGL(gs::sh_fwd_kernel<4>, grid, block, 0, s, n, K, dirs, coeffs, masks, out); break;
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

#define GS_SH_BWD_CASE(D)                                                                                  \
  case D:                                                                                                  \
    if (v_dirs)                                                                                            \
      hipLaunchKernelGGL((gs::sh_bwd_kernel<D, true>), grid, block, 0, s, n, K, dirs, coeffs, masks, v_out, \
                         v_coeffs, v_dirs);                                                                \
    else                                                                                                   \
      hipLaunchKernelGGL((gs::sh_bwd_kernel<D, false>), grid, block, 0, s, n, K, dirs, coeffs, masks,      \
                         v_out, v_coeffs, v_dirs);                                                         \
    break;

extern "C" int gs_sh_bwd(int64_t n, int32_t K, int32_t degree, const float* dirs, const float* coeffs,
                         const uint8_t* masks, const float* v_out, float* v_coeffs, float* v_dirs,
                         void* stream) {
  if (n < 0 || degree < 0 || degree > 4 || (degree + 1) * (degree + 1) > K) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!dirs || !coeffs || !v_out || !v_coeffs) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(n, 256)), block(256);
  switch (degree) {
    GS_SH_BWD_CASE(0)
    GS_SH_BWD_CASE(1)
    GS_SH_BWD_CASE(2)
    GS_SH_BWD_CASE(3)
    default:
      GS_SH_BWD_CASE(4)
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}
