// Standalone spherical_harmonics forward/backward (HBM-bound: 12+12K B read, 12 B written per row).
// One thread per (flattened) Gaussian row; each lane streams its own contiguous K*12-byte
// coefficient block, so every fetched line is fully consumed.
#include "sh_math.hpp"

namespace gs {

template <int DEG>
__global__ __launch_bounds__(256) void sh_fwd_kernel(int64_t n, int K, const float* __restrict__ dirs,
                                                     const float* __restrict__ coeffs,
                                                     const uint8_t* __restrict__ masks,
                                                     float* __restrict__ out) {
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
  constexpr int NB = (DEG + 1) * (DEG + 1);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc[NB * 3];
#pragma unroll
  for (int k = 0; k < NB * 3; ++k) acc[k] = 0.f;
  float vx = 0.f, vy = 0.f, vz = 0.f;
  if (masks == nullptr || masks[i]) {
    const float v[3] = {v_out[3 * i], v_out[3 * i + 1], v_out[3 * i + 2]};
    sh_eval_vjp<DEG, DIRGRAD>(coeffs + i * K * 3, dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], v, acc, vx,
                              vy, vz);
  }
  float* vc = v_coeffs + i * K * 3;
#pragma unroll
  for (int k = 0; k < NB * 3; ++k) vc[k] = acc[k];
  for (int k = NB * 3; k < K * 3; ++k) vc[k] = 0.f;
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
