// The callers' per-step pre-pass fused into one kernel per Gaussian class (training
// half).  Before every rasterization() the reference turns raw Parameters into render inputs with a chain of torch
// ops (/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:359-386 `get_gaussians`):
//   colors   = cat(features_dc[:, None, :], features_rest)                      (a [N,K,3] copy)
//   viewdirs = normalize(means.detach() - camtoworld[:3, 3])
//   rgbs     = clamp(spherical_harmonics(n, viewdirs, colors) + 0.5, 0, 1)      (or sigmoid(features_dc), sh_degree 0)
//   opacities = sigmoid(_opacities) [* valid mask, nodes/rigid.py:493]; scales = exp(_scales) (ball / 2-D variants,
//   vanilla.py:112-128); quats = _quats / |_quats|
// followed by ten isnan/isinf reductions that each synchronise the host (vanilla.py:388-393) and a torch.cat of every
// class's five arrays (models/trainers/base.py:365-366).  Here each class's kernel reads the raw rows once and writes
// its slice of the scene-level arrays directly; non-finite outputs raise bits in a device flag word instead of
// stalling the host.  One thread per Gaussian; HBM-bound: 12K + 44 B read, 56 B + 1 B (clamp mask) written per row,
// the backward reads 44 + 44 B and writes 12K + 32 B (dense `v_features_rest` rows through the wave-row LDS transpose).
#include "sh_math.hpp"

namespace gs {

// bits of the finite-check word: (key index 0.._means,1.._opacities,2.._rgbs,3.._scales,4.._quats) * 2 + (0 NaN, 1 Inf)
__device__ __forceinline__ uint32_t bad_bits(float v, int key) {
  const uint32_t u = __float_as_uint(v) & 0x7fffffffu;
  return u > 0x7f800000u ? (1u << (2 * key)) : (u == 0x7f800000u ? (2u << (2 * key)) : 0u);
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

enum { SCALE_3D = 0, SCALE_BALL = 1, SCALE_2D = 2 };

// DEG < 0: colour = sigmoid(features_dc) (the class's sh_degree is 0, vanilla.py:371-372)
template <int DEG>
__global__ __launch_bounds__(256) void activate_fwd_kernel(
    int64_t n, int KR, const float* __restrict__ means, const float* __restrict__ cam_center,
    const float* __restrict__ dc, const float* __restrict__ rest, const float* __restrict__ raw_opac,
    const float* __restrict__ opac_mask, const float* __restrict__ raw_scales, int scale_mode,
    const float* __restrict__ raw_quats, float* __restrict__ o_means, float* __restrict__ o_opac,
    float* __restrict__ o_rgbs, float* __restrict__ o_scales, float* __restrict__ o_quats,
    uint8_t* __restrict__ clamp_mask, uint32_t* __restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t bad = 0;
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  o_means[3 * i] = mx;
  o_means[3 * i + 1] = my;
  o_means[3 * i + 2] = mz;
  bad |= bad_bits(mx, 0) | bad_bits(my, 0) | bad_bits(mz, 0);

  float rgb[3];
  uint8_t cm = 0;
  if constexpr (DEG < 0) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) rgb[ch] = sigmoidf(dc[3 * i + ch]);
  } else {
    constexpr int NB = (DEG + 1) * (DEG + 1);
    // viewdirs are normalised by the caller and again inside spherical_harmonics: same value to an ulp
    float dx = mx - cam_center[0], dy = my - cam_center[1], dz = mz - cam_center[2];
    const float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
    dx *= inorm; dy *= inorm; dz *= inorm;
    const float in2 = rsqrtf(dx * dx + dy * dy + dz * dz);
    float b[NB];
    sh_basis<DEG>(dx * in2, dy * in2, dz * in2, b);
    const float* r = rest + i * (int64_t)KR * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      float a = b[0] * dc[3 * i + ch];
#pragma unroll
      for (int k = 1; k < NB; ++k) a = fmaf(b[k], r[3 * (k - 1) + ch], a);
      a += 0.5f;
      // torch.clamp passes the gradient where 0 <= x <= 1 (NaN: no gradient, and the flag below reports it)
      if (a >= 0.f && a <= 1.f) cm |= (uint8_t)(1u << ch);
      rgb[ch] = a != a ? a : fminf(fmaxf(a, 0.f), 1.f);
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    o_rgbs[3 * i + ch] = rgb[ch];
    bad |= bad_bits(rgb[ch], 2);
  }
  clamp_mask[i] = cm;

  float o = sigmoidf(raw_opac[i]);
  if (opac_mask) o *= opac_mask[i];
  o_opac[i] = o;
  bad |= bad_bits(o, 1);

  float s[3];
  if (scale_mode == SCALE_BALL) {
    s[0] = s[1] = s[2] = expf(raw_scales[i]);
  } else {
    s[0] = expf(raw_scales[3 * i]);
    s[1] = expf(raw_scales[3 * i + 1]);
    s[2] = scale_mode == SCALE_2D ? 0.f : expf(raw_scales[3 * i + 2]);
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    o_scales[3 * i + a] = s[a];
    bad |= bad_bits(s[a], 3);
  }

  const float4 q = reinterpret_cast<const float4*>(raw_quats)[i];
  const float qn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  const float4 qo = make_float4(q.x / qn, q.y / qn, q.z / qn, q.w / qn);
  reinterpret_cast<float4*>(o_quats)[i] = qo;
  bad |= bad_bits(qo.x, 4) | bad_bits(qo.y, 4) | bad_bits(qo.z, 4) | bad_bits(qo.w, 4);

  if (bad) atomicOr(flag, bad);
}

template <int DEG>
struct RestTile {
  static constexpr int NB = DEG < 0 ? 1 : (DEG + 1) * (DEG + 1);
  static constexpr int NR3 = (NB - 1) * 3;            // floats of a v_features_rest row that can be non-zero
  static constexpr int ROWS = (NR3 <= 48) ? 64 : 32;  // rows per round
  static constexpr int FLOATS = ROWS * (NR3 + 1);
};

template <int DEG>
__global__ __launch_bounds__(256) void activate_bwd_kernel(
    int64_t n, int KR, const float* __restrict__ means, const float* __restrict__ cam_center,
    const float* __restrict__ dc, const float* __restrict__ raw_opac, const float* __restrict__ opac_mask,
    const float* __restrict__ raw_scales, int scale_mode, const float* __restrict__ raw_quats,
    const uint8_t* __restrict__ clamp_mask, const float* __restrict__ v_opac, const float* __restrict__ v_rgbs,
    const float* __restrict__ v_scales, const float* __restrict__ v_quats, float* __restrict__ g_dc,
    float* __restrict__ g_rest, float* __restrict__ g_opac, float* __restrict__ g_scales,
    float* __restrict__ g_quats) {
  using TL = RestTile<DEG>;
  constexpr int NB = TL::NB;
  __shared__ float s_t[4 * TL::FLOATS];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool live = i < n;

  float vr[3] = {0.f, 0.f, 0.f};
  float b[NB];
  b[0] = 0.f;
  if (live) {
    if constexpr (DEG < 0) {
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const float sg = sigmoidf(dc[3 * i + ch]);
        g_dc[3 * i + ch] = v_rgbs[3 * i + ch] * sg * (1.f - sg);
      }
    } else {
      const uint8_t cm = clamp_mask[i];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) vr[ch] = ((cm >> ch) & 1) ? v_rgbs[3 * i + ch] : 0.f;
      float dx = means[3 * i] - cam_center[0], dy = means[3 * i + 1] - cam_center[1],
            dz = means[3 * i + 2] - cam_center[2];
      const float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
      dx *= inorm; dy *= inorm; dz *= inorm;
      const float in2 = rsqrtf(dx * dx + dy * dy + dz * dz);
      sh_basis<DEG>(dx * in2, dy * in2, dz * in2, b);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) g_dc[3 * i + ch] = b[0] * vr[ch];
    }
  }
  // v_features_rest [n, KR, 3]: dense rows, bands above the degree in use are zeros
  if (KR > 0) {
    float* t = s_t + wave * TL::FLOATS;
    const int64_t row_w = i - lane;
#pragma unroll 1
    for (int h = 0; h < 64 / TL::ROWS; ++h) {
      if constexpr (NB > 1) {
        if (lane / TL::ROWS == h) {
#pragma unroll
          for (int k = 1; k < NB; ++k)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
              t[(lane % TL::ROWS) * (TL::NR3 + 1) + 3 * (k - 1) + ch] = live ? b[k] * vr[ch] : 0.f;
        }
      }
      const int64_t row0 = row_w + h * TL::ROWS;
      const int rows = (int)min((int64_t)TL::ROWS, n - row0);
      if (rows <= 0) continue;
      if constexpr (TL::NR3 == 0) {
        float* base = g_rest + row0 * KR * 3;
        for (int m = lane; m < rows * KR * 3; m += 64) base[m] = 0.f;
      } else {
        wave_rows_store<TL::NR3>(t, g_rest + row0 * KR * 3, rows, KR * 3, lane);
      }
    }
  }
  if (!live) return;

  {
    const float o = sigmoidf(raw_opac[i]);
    float v = v_opac[i];
    if (opac_mask) v *= opac_mask[i];
    g_opac[i] = v * o * (1.f - o);
  }
  if (scale_mode == SCALE_BALL) {
    g_scales[i] = (v_scales[3 * i] + v_scales[3 * i + 1] + v_scales[3 * i + 2]) * expf(raw_scales[i]);
  } else {
    g_scales[3 * i] = v_scales[3 * i] * expf(raw_scales[3 * i]);
    g_scales[3 * i + 1] = v_scales[3 * i + 1] * expf(raw_scales[3 * i + 1]);
    g_scales[3 * i + 2] = scale_mode == SCALE_2D ? 0.f : v_scales[3 * i + 2] * expf(raw_scales[3 * i + 2]);
  }
  {
    // q_hat = q / |q|:  v_q = (v - q_hat (q_hat . v)) / |q|
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[i];
    const float4 v = reinterpret_cast<const float4*>(v_quats)[i];
    const float inv = 1.f / sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float hx = q.x * inv, hy = q.y * inv, hz = q.z * inv, hw = q.w * inv;
    const float d = hx * v.x + hy * v.y + hz * v.z + hw * v.w;
    reinterpret_cast<float4*>(g_quats)[i] =
        make_float4((v.x - hx * d) * inv, (v.y - hy * d) * inv, (v.z - hz * d) * inv, (v.w - hw * d) * inv);
  }
}

}  // namespace gs

// degree: SH degree in use (0..4), or -1 for colour = sigmoid(features_dc).  K_rest = rows of features_rest per
// Gaussian (K - 1); (degree + 1)^2 - 1 <= K_rest.  scale_mode 0: raw_scales [n,3]; 1: [n,1] repeated (ball); 2: [n,3]
// with the third axis forced to 0 (2-D Gaussians).  Outputs are this class's slices of the scene-level arrays.
extern "C" int gs_activate_fwd(int64_t n, int32_t K_rest, int32_t degree, const float* means, const float* cam_center,
                               const float* features_dc, const float* features_rest, const float* raw_opacities,
                               const float* opacity_mask, const float* raw_scales, int32_t scale_mode,
                               const float* raw_quats, float* out_means, float* out_opacities, float* out_rgbs,
                               float* out_scales, float* out_quats, uint8_t* clamp_mask, uint32_t* finite_flag,
                               void* stream) {
  if (n < 0 || K_rest < 0 || degree < -1 || degree > 4 || scale_mode < 0 || scale_mode > 2) return GS_ERR_INVALID_ARG;
  if (degree >= 0 && (degree + 1) * (degree + 1) - 1 > K_rest) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!means || !features_dc || !raw_opacities || !raw_scales || !raw_quats || !out_means || !out_opacities ||
      !out_rgbs || !out_scales || !out_quats || !clamp_mask || !finite_flag)
    return GS_ERR_INVALID_ARG;
  if (degree >= 0 && !cam_center) return GS_ERR_INVALID_ARG;
  if (degree >= 1 && !features_rest) return GS_ERR_INVALID_ARG;
  if (((uintptr_t)raw_quats | (uintptr_t)out_quats) & 15) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(n, 256)), block(256);
#define GS_ACT_FWD(D)                                                                                              \
  hipLaunchKernelGGL(gs::activate_fwd_kernel<D>, grid, block, 0, s, n, K_rest, means, cam_center, features_dc,      \
                     features_rest, raw_opacities, opacity_mask, raw_scales, scale_mode, raw_quats, out_means,      \
                     out_opacities, out_rgbs, out_scales, out_quats, clamp_mask, finite_flag)
  switch (degree) {
    case -1: GS_ACT_FWD(-1); break;
    case 0: GS_ACT_FWD(0); break;
    case 1: GS_ACT_FWD(1); break;
    case 2: GS_ACT_FWD(2); break;
    case 3: GS_ACT_FWD(3); break;
    default: GS_ACT_FWD(4); break;
  }
#undef GS_ACT_FWD
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// Gradients of the raw Parameters from the gradients of this class's slices.  `v_means` needs no kernel (the means
// pass through; view directions are detached, vanilla.py:366).  g_features_rest [n,K_rest,3] is written whole.
extern "C" int gs_activate_bwd(int64_t n, int32_t K_rest, int32_t degree, const float* means, const float* cam_center,
                               const float* features_dc, const float* raw_opacities, const float* opacity_mask,
                               const float* raw_scales, int32_t scale_mode, const float* raw_quats,
                               const uint8_t* clamp_mask, const float* v_opacities, const float* v_rgbs,
                               const float* v_scales, const float* v_quats, float* g_features_dc,
                               float* g_features_rest, float* g_raw_opacities, float* g_raw_scales, float* g_raw_quats,
                               void* stream) {
  if (n < 0 || K_rest < 0 || degree < -1 || degree > 4 || scale_mode < 0 || scale_mode > 2) return GS_ERR_INVALID_ARG;
  if (degree >= 0 && (degree + 1) * (degree + 1) - 1 > K_rest) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!means || !features_dc || !raw_opacities || !raw_scales || !raw_quats || !clamp_mask || !v_opacities ||
      !v_rgbs || !v_scales || !v_quats || !g_features_dc || !g_raw_opacities || !g_raw_scales || !g_raw_quats)
    return GS_ERR_INVALID_ARG;
  if (degree >= 0 && !cam_center) return GS_ERR_INVALID_ARG;
  if (K_rest > 0 && !g_features_rest) return GS_ERR_INVALID_ARG;
  if (((uintptr_t)raw_quats | (uintptr_t)v_quats | (uintptr_t)g_raw_quats) & 15) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(n, 256)), block(256);
#define GS_ACT_BWD(D)                                                                                              \
  hipLaunchKernelGGL(gs::activate_bwd_kernel<D>, grid, block, 0, s, n, K_rest, means, cam_center, features_dc,      \
                     raw_opacities, opacity_mask, raw_scales, scale_mode, raw_quats, clamp_mask, v_opacities, v_rgbs, \
                     v_scales, v_quats, g_features_dc, g_features_rest, g_raw_opacities, g_raw_scales, g_raw_quats)
  switch (degree) {
    case -1: GS_ACT_BWD(-1); break;
    case 0: GS_ACT_BWD(0); break;
    case 1: GS_ACT_BWD(1); break;
    case 2: GS_ACT_BWD(2); break;
    case 3: GS_ACT_BWD(3); break;
    default: GS_ACT_BWD(4); break;
  }
#undef GS_ACT_BWD
  GS_CHECK_LAUNCH();
  return GS_OK;
}
