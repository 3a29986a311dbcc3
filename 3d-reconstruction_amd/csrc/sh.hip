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
  // each lane streams its own row: the read side is already at what a pure-read stream reaches on this part
  // (3.7 TB/s here; a torch reduction over the same bytes 3.8), and with fewer bands in use than K a lane only
  // touches the lines it needs -- the cooperative LDS route (measured) gains nothing at degree 3 and loses below it
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float o[3] = {0.f, 0.f, 0.f};
  if (masks == nullptr || masks[i]) {
    sh_eval<DEG>(coeffs + i * K * 3, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], o);
  }
  out[3 * i] = o[0];
  out[3 * i + 1] = o[1];
  out[3 * i + 2] = o[2];
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

}  // namespace gs

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
