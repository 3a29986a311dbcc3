// RGB loss terms next to the rasterizer's output, fused.
// The reference computes, on the [H,W,3] render and ground truth (both first multiplied by a [H,W] validity mask),
//   Ll1     = |gt - pred|.mean()                                   (models/trainers/base.py:540)
//   simloss = 1 - SSIM(data_range=1, size_average=True, channel=3)(gt, pred)          (base.py:114, :541)
// with pytorch_msssim 1.0.0 (requirements.txt:7; not in /root/reference, restated from its published algorithm): an
// 11-tap Gaussian window (sigma 1.5, normalised), separable VALID convolutions (no padding: the SSIM map is
// (H-10) x (W-10)), per pixel and channel
//   m = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),  C1 = 0.01^2, C2 = 0.03^2,
// and the mean of m over pixels and channels.  In torch that is ten depthwise convolutions and ~25 elementwise passes
// forward, twice that backward.  Here: one forward kernel (tile in LDS, horizontal then vertical filter of the five
// moments x, y, x^2, y^2, xy, the SSIM value, the three partial derivatives the backward needs, the L1 term, per-block
// partial sums), a one-block deterministic reduction, and one backward kernel (transposed filter of the derivative
// maps + the L1 sign + the mask).  HBM-bound: forward reads 2 x 12 B + 4 B and writes 36 B per pixel, backward reads
// 36 + 28 B and writes 12 B.
#include "gs_common.hpp"

namespace gs {

constexpr int SS_WIN = 11, SS_HALO = SS_WIN - 1;
constexpr int SS_T = 16;                 // output tile edge
constexpr int SS_IN = SS_T + SS_HALO;    // 26 input rows / columns per tile
constexpr float SS_C1 = 0.01f * 0.01f, SS_C2 = 0.03f * 0.03f;

struct SsimWin { float w[SS_WIN]; };

// block (bx, by) owns the pixels [by*16, +16) x [bx*16, +16): their L1 terms, and the SSIM map entries with those
// indices (map entry (oy, ox) covers input rows oy..oy+10, columns ox..ox+10; valid for oy < H-10, ox < W-10)
__global__ __launch_bounds__(256) void l1_ssim_fwd_kernel(int H, int W, SsimWin win, const float* __restrict__ pred,
                                                          const float* __restrict__ gt, const float* __restrict__ mask,
                                                          float* __restrict__ maps, float* __restrict__ partials) {
  __shared__ float sx[SS_IN][SS_IN * 3 + 1], sy[SS_IN][SS_IN * 3 + 1];
  __shared__ float hb[5][SS_IN][SS_T * 3 + 1];
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int y0 = blockIdx.y * SS_T, x0 = blockIdx.x * SS_T;
  const int Ho = H - SS_HALO, Wo = W - SS_HALO;
  float l1 = 0.f;
  // load the 26 x 26 x 3 tile of both images (masked), zero outside the image
  for (int i = tid; i < SS_IN * SS_IN; i += 256) {
    const int r = i / SS_IN, c = i - r * SS_IN;
    const int y = y0 + r, x = x0 + c;
    float p[3] = {0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
    if (y < H && x < W) {
      const int64_t o = ((int64_t)y * W + x) * 3;
      const float mk = mask ? mask[(int64_t)y * W + x] : 1.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) { p[k] = pred[o + k] * mk; g[k] = gt[o + k] * mk; }
      if (r < SS_T && c < SS_T) l1 += fabsf(g[0] - p[0]) + fabsf(g[1] - p[1]) + fabsf(g[2] - p[2]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { sx[r][c * 3 + k] = p[k]; sy[r][c * 3 + k] = g[k]; }
  }
  __syncthreads();
  // horizontal filter of the five moments: 26 rows x 16 columns x 3 channels
  for (int i = tid; i < SS_IN * SS_T * 3; i += 256) {
    const int r = i / (SS_T * 3), j = i - r * (SS_T * 3);  // j = column * 3 + channel
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
    for (int t = 0; t < SS_WIN; ++t) {
      const float x = sx[r][j + 3 * t], y = sy[r][j + 3 * t], w = win.w[t];
      a0 = fmaf(w, x, a0); a1 = fmaf(w, y, a1);
      a2 = fmaf(w, x * x, a2); a3 = fmaf(w, y * y, a3); a4 = fmaf(w, x * y, a4);
    }
    hb[0][r][j] = a0; hb[1][r][j] = a1; hb[2][r][j] = a2; hb[3][r][j] = a3; hb[4][r][j] = a4;
  }
  __syncthreads();
  // vertical filter + SSIM: one thread per output pixel, three channels
  float ss = 0.f;
  {
    const int r = tid >> 4, c = tid & 15;
    const int oy = y0 + r, ox = x0 + c;
    if (oy < Ho && ox < Wo) {
      float dm[9];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
        for (int t = 0; t < SS_WIN; ++t) {
          const float w = win.w[t];
          mu1 = fmaf(w, hb[0][r + t][c * 3 + k], mu1);
          mu2 = fmaf(w, hb[1][r + t][c * 3 + k], mu2);
          e11 = fmaf(w, hb[2][r + t][c * 3 + k], e11);
          e22 = fmaf(w, hb[3][r + t][c * 3 + k], e22);
          e12 = fmaf(w, hb[4][r + t][c * 3 + k], e12);
        }
        const float s1 = e11 - mu1 * mu1, s2 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
        const float A = mu1 * mu1 + mu2 * mu2 + SS_C1, Bq = s1 + s2 + SS_C2;
        const float Cq = 2.f * mu1 * mu2 + SS_C1, D = 2.f * s12 + SS_C2;
        const float iA = 1.f / A, iB = 1.f / Bq;
        const float m = Cq * D * iA * iB;
        ss += m;
        // partial derivatives w.r.t. the PRED-side moments (mu1 with E[x^2], E[xy] held fixed; s1; s12)
        dm[k] = 2.f * iA * iB * (mu2 * (D - Cq) - mu1 * Cq * D * iA + mu1 * Cq * D * iB);
        dm[3 + k] = -m * iB;
        dm[6 + k] = 2.f * Cq * iA * iB;
      }
      if (maps) {
        float* o = maps + ((int64_t)oy * Wo + ox) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = dm[k];
      }
    }
  }
  // block partial sums (fixed order: deterministic)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { l1 += __shfl_down(l1, off); ss += __shfl_down(ss, off); }
  if (lane == 0) { red[0][wave] = l1; red[1][wave] = ss; }
  __syncthreads();
  if (tid == 0) {
    const int64_t b = (int64_t)blockIdx.y * gridDim.x + blockIdx.x, nb = (int64_t)gridDim.x * gridDim.y;
    partials[b] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    partials[nb + b] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// out[0] = sum(partials[0..nb)) * s0, out[1] = sum(partials[nb..2nb)) * s1; one workgroup, fixed summation order
__global__ __launch_bounds__(1024) void l1_ssim_reduce_kernel(int64_t nb, const float* __restrict__ partials, double s0,
                                                              double s1, float* __restrict__ out) {
  __shared__ double red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a = 0.0, b = 0.0;
  for (int64_t i = tid; i < nb; i += 1024) { a += (double)partials[i]; b += (double)partials[nb + i]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); }
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
  __syncthreads();
  if (tid == 0) {
    double ta = 0.0, tb = 0.0;
    for (int w = 0; w < 16; ++w) { ta += red[0][w]; tb += red[1][w]; }
    out[0] = (float)(ta * s0);
    out[1] = (float)(tb * s1);
  }
}

// v_pred[p] = mask[p] * ( g_l1 * sign(pred_m - gt_m) / (3HW)
//                         + g_ssim / (3 Ho Wo) * sum_{o in window of p} w(o,p) (dmu1[o] + 2 x[p] ds1[o] + y[p] ds12[o]) )
__global__ __launch_bounds__(256) void l1_ssim_bwd_kernel(int H, int W, SsimWin win, const float* __restrict__ pred,
                                                          const float* __restrict__ gt, const float* __restrict__ mask,
                                                          const float* __restrict__ maps, const float* __restrict__ g_out,
                                                          float s_l1, float s_ssim, float* __restrict__ v_pred) {
  // map tile: entries (oy, ox) with oy in [y0-10, y0+16), ox in [x0-10, x0+16): 26 x 26 x 9 floats
  __shared__ float sm[SS_IN][SS_IN * 9 + 1];
  __shared__ float hb[SS_IN][SS_T * 9 + 1];
  const int tid = threadIdx.x;
  const int y0 = blockIdx.y * SS_T, x0 = blockIdx.x * SS_T;
  const int Ho = H - SS_HALO, Wo = W - SS_HALO;
  for (int i = tid; i < SS_IN * SS_IN * 9; i += 256) {
    const int r = i / (SS_IN * 9), j = i - r * (SS_IN * 9);
    const int c = j / 9, k = j - c * 9;
    const int oy = y0 - SS_HALO + r, ox = x0 - SS_HALO + c;
    sm[r][j] = (oy >= 0 && oy < Ho && ox >= 0 && ox < Wo) ? maps[((int64_t)oy * Wo + ox) * 9 + k] : 0.f;
  }
  __syncthreads();
  // horizontal: input column x receives from map columns ox = x - t (t = 0..10) with weight w[t]
  for (int i = tid; i < SS_IN * SS_T * 9; i += 256) {
    const int r = i / (SS_T * 9), j = i - r * (SS_T * 9);
    const int c = j / 9, k = j - c * 9;
    float a = 0.f;
#pragma unroll
    for (int t = 0; t < SS_WIN; ++t) a = fmaf(win.w[t], sm[r][(c + SS_HALO - t) * 9 + k], a);
    hb[r][j] = a;
  }
  __syncthreads();
  const int r = tid >> 4, c = tid & 15;
  const int y = y0 + r, x = x0 + c;
  if (y >= H || x >= W) return;
  const float mk = mask ? mask[(int64_t)y * W + x] : 1.f;
  const float gl1 = g_out[0] * s_l1, gss = g_out[1] * s_ssim;
  const int64_t o = ((int64_t)y * W + x) * 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float d0 = 0.f, d1 = 0.f, d2 = 0.f;
#pragma unroll
    for (int t = 0; t < SS_WIN; ++t) {
      const float w = win.w[t];
      d0 = fmaf(w, hb[r + SS_HALO - t][c * 9 + k], d0);
      d1 = fmaf(w, hb[r + SS_HALO - t][c * 9 + 3 + k], d1);
      d2 = fmaf(w, hb[r + SS_HALO - t][c * 9 + 6 + k], d2);
    }
    const float p = pred[o + k] * mk, g = gt[o + k] * mk;
    const float diff = p - g;
    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);  // torch.abs has gradient 0 at 0
    v_pred[o + k] = mk * (gl1 * sgn + gss * (d0 + 2.f * p * d1 + g * d2));
  }
}

static SsimWin make_window() {
  SsimWin w;
  // pytorch_msssim builds the window in fp32: exp(-(i - 5)^2 / (2 * 1.5^2)), then a division by the fp32 sum
  float gf[SS_WIN], sf = 0.f;
  for (int i = 0; i < SS_WIN; ++i) {
    const double c = (double)(i - SS_WIN / 2);
    gf[i] = (float)exp(-(c * c) / (2.0 * 1.5 * 1.5));
    sf += gf[i];
  }
  for (int i = 0; i < SS_WIN; ++i) w.w[i] = gf[i] / sf;
  return w;
}

}  // namespace gs

extern "C" int64_t gs_l1_ssim_workspace_floats(int32_t H, int32_t W) {
  if (H < 1 || W < 1) return 0;
  return 2 * gs::ceil_div((int64_t)H, gs::SS_T) * gs::ceil_div((int64_t)W, gs::SS_T);
}

// out[0] = mean |gt*mask - pred*mask| over H*W*3, out[1] = SSIM(gt*mask, pred*mask) (mean of the (H-10)x(W-10)x3 map).
// maps (NULL when no gradient is needed): [(H-10)*(W-10)*9] floats kept for gs_l1_ssim_bwd.
extern "C" int gs_l1_ssim_fwd(int32_t H, int32_t W, const float* pred, const float* gt, const float* mask, float* maps,
                              float* workspace, int64_t workspace_floats, float* out, void* stream) {
  if (H <= gs::SS_HALO || W <= gs::SS_HALO) return GS_ERR_UNSUPPORTED;  // pytorch_msssim skips the filter there
  if (!pred || !gt || !workspace || !out) return GS_ERR_INVALID_ARG;
  if (workspace_floats < gs_l1_ssim_workspace_floats(H, W)) return GS_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid((unsigned)gs::ceil_div(W, gs::SS_T), (unsigned)gs::ceil_div(H, gs::SS_T));
  const int64_t nb = (int64_t)grid.x * grid.y;
  hipLaunchKernelGGL(gs::l1_ssim_fwd_kernel, grid, dim3(256), 0, s, H, W, gs::make_window(), pred, gt, mask, maps,
                     workspace);
  hipLaunchKernelGGL(gs::l1_ssim_reduce_kernel, dim3(1), dim3(1024), 0, s, nb, workspace,
                     1.0 / ((double)H * W * 3.0), 1.0 / ((double)(H - gs::SS_HALO) * (W - gs::SS_HALO) * 3.0), out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// v_pred [H,W,3] = g_out[0] * dL1/dpred + g_out[1] * dSSIM/dpred (g_out: 2 floats on the device).
extern "C" int gs_l1_ssim_bwd(int32_t H, int32_t W, const float* pred, const float* gt, const float* mask,
                              const float* maps, const float* g_out, float* v_pred, void* stream) {
  if (H <= gs::SS_HALO || W <= gs::SS_HALO) return GS_ERR_UNSUPPORTED;
  if (!pred || !gt || !maps || !g_out || !v_pred) return GS_ERR_INVALID_ARG;
  const dim3 grid((unsigned)gs::ceil_div(W, gs::SS_T), (unsigned)gs::ceil_div(H, gs::SS_T));
  hipLaunchKernelGGL(gs::l1_ssim_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, H, W, gs::make_window(), pred,
                     gt, mask, maps, g_out, (float)(1.0 / ((double)H * W * 3.0)),
                     (float)(1.0 / ((double)(H - gs::SS_HALO) * (W - gs::SS_HALO) * 3.0)), v_pred);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The per-pixel scalar terms of compute_losses in their shipped configuration (R/configs/omnire.yaml:24-31):
//   sky / mask loss   binary_cross_entropy(opacity * valid, (1 - sky_mask) * valid).mean()        (base.py:536-550,
//                     models/losses.py:86-88; torch clamps both logs at -100)
//   depth loss        DepthLoss("l1", normalize=False, inverse=False, "mean_on_hit") (losses.py:96-199, base.py:553-557):
//                     hit = (gt > 0) * valid; mean of |pred*hit - gt*hit| over {gt*hit in (0.01, 80), pred*hit > 1e-4}
//   opacity entropy   (-p log p).mean(), p = clamp(opacity, 1e-6, 1 - 1e-6)                       (base.py:568-573)
// One pass over the pixels -> per-workgroup partial sums {bce, depth, depth count, entropy} -> fixed-order reduction.
// The torch version is ~30 elementwise kernels, a boolean-mask gather (with its host synchronisation) and four
// reductions.
namespace gs {

struct PixelTerms { float bce, dabs, dcnt, ent; };

// The configurable variants of the reference's mask and depth losses (gs_pixel_loss_opts, include/gsplat_mi355x.h).
struct DepthTerm { bool sel; float loss, dloss_dpred; };

__device__ __forceinline__ DepthTerm depth_term(const gs_pixel_loss_opts& o, float pred, float gt, float vm,
                                                const uint8_t* select, int64_t i) {
  // DepthLoss._compute_depth_loss (models/losses.py:113-151): hit = (gt > 0) * valid multiplies both maps; a pixel is
  // selected when gt*hit in (0.01, max_depth) and pred*hit > 1e-4; then safe_normalize_depth, 1/x, and the loss type
  DepthTerm t = {false, 0.f, 0.f};
  const float hit = (gt > 0.f ? 1.f : 0.f) * vm;
  const float pd = pred * hit, gd = gt * hit;
  if (!(gd > 0.01f && gd < o.max_depth && pd > 0.0001f)) return t;
  if (select && !select[i]) return t;  // percentile clipping: the caller's keep-mask over the selected pixels
  float p = pd, g = gd, dp = hit;  // dp = d p / d pred
  if (o.depth_normalize) {
    const float q = pd / o.max_depth;
    p = fminf(fmaxf(q, 1e-6f), 1.f);
    g = fminf(fmaxf(gd / o.max_depth, 1e-6f), 1.f);
    dp = (q >= 1e-6f && q <= 1.f) ? hit / o.max_depth : 0.f;  // torch.clamp passes the gradient on [min, max]
  }
  if (o.depth_inverse) {
    dp *= -1.f / (p * p);
    p = 1.f / p;
    g = 1.f / g;
  }
  const float d = p - g, ad = fabsf(d), sg = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
  float dl;
  if (o.depth_type == 1) { t.loss = d * d; dl = 2.f * d; }                                  // F.mse_loss
  else if (o.depth_type == 2) { t.loss = ad < 1.f ? 0.5f * d * d : ad - 0.5f; dl = ad < 1.f ? d : sg; }  // smooth_l1, beta 1
  else { t.loss = ad; dl = sg; }                                                            // F.l1_loss
  t.sel = true;
  t.dloss_dpred = dl * dp;
  return t;
}

__device__ __forceinline__ PixelTerms pixel_terms(const gs_pixel_loss_opts& o, int64_t i, const float* opacity,
                                                  const float* depth, const float* sky, const float* valid,
                                                  const float* lidar, const uint8_t* select, float* err_map) {
  PixelTerms t = {0.f, 0.f, 0.f, 0.f};
  const float vm = valid ? valid[i] : 1.f, op = opacity[i];
  if (sky) {
    const float p = op * vm, g = (1.f - sky[i]) * vm;
    if (o.bce_mode == 1) {
      // SafeBCE.forward (models/losses.py:41-54): -where(y == 0, log(1 - x), log(x)).clamp_min(ln limit), x, y in [0, 1]
      const float x = fminf(fmaxf(p, 0.f), 1.f), y = fminf(fmaxf(g, 0.f), 1.f), ln_limit = logf(o.bce_limit);
      t.bce = -fmaxf(y == 0.f ? logf(1.f - x) : logf(x), ln_limit);
    } else {
      t.bce = -(g * fmaxf(logf(p), -100.f) + (1.f - g) * fmaxf(logf(1.f - p), -100.f));
    }
  }
  if (lidar) {
    const DepthTerm d = depth_term(o, depth[i], lidar[i], vm, select, i);
    if (d.sel) { t.dabs = d.loss; t.dcnt = 1.f; }
    if (err_map) err_map[i] = d.sel ? d.loss : -1.f;
  }
  const float pc = fminf(fmaxf(op, 1e-6f), 1.f - 1e-6f);
  t.ent = -pc * logf(pc);
  return t;
}

__global__ __launch_bounds__(256) void pixel_losses_fwd_kernel(gs_pixel_loss_opts o, int64_t n,
                                                               const float* __restrict__ opacity,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ sky,
                                                               const float* __restrict__ valid,
                                                               const float* __restrict__ lidar,
                                                               const uint8_t* __restrict__ select,
                                                               float* __restrict__ err_map,
                                                               float* __restrict__ partials) {
  __shared__ float red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const PixelTerms t = pixel_terms(o, i, opacity, depth, sky, valid, lidar, select, err_map);
    a[0] += t.bce; a[1] += t.dabs; a[2] += t.dcnt; a[3] += t.ent;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    if (lane == 0) red[k][wave] = a[k];
  }
  __syncthreads();
  if (tid < 4) partials[(int64_t)tid * gridDim.x + blockIdx.x] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// depth-term denominator by reduction: mean_on_hit -> selected pixels, mean_on_hw -> all pixels, sum -> 1
__device__ __forceinline__ double depth_denominator(int reduction, double count, int64_t n) {
  return reduction == 1 ? (double)n : (reduction == 2 ? 1.0 : count);
}

// out = {bce mean, depth term, number of selected pixels, entropy mean}
__global__ __launch_bounds__(256) void pixel_losses_reduce_kernel(int nb, int64_t n, int depth_reduction,
                                                                  const float* __restrict__ partials,
                                                                  float* __restrict__ out) {
  __shared__ double red[4][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[4] = {0.0, 0.0, 0.0, 0.0};
  for (int i = tid; i < nb; i += 256)
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] += (double)partials[(int64_t)k * nb + i];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    if (lane == 0) red[k][wave] = a[k];
  }
  __syncthreads();
  if (tid == 0) {
    double s[4];
    for (int k = 0; k < 4; ++k) s[k] = (red[k][0] + red[k][1]) + (red[k][2] + red[k][3]);
    out[0] = (float)(s[0] / (double)n);
    // mean_on_hit with no selected pixel: 0 / 0 = NaN, as torch's mean of an empty tensor
    out[1] = (float)(s[1] / depth_denominator(depth_reduction, s[2], n));
    out[2] = (float)s[2];
    out[3] = (float)(s[3] / (double)n);
  }
}

__global__ __launch_bounds__(256) void pixel_losses_bwd_kernel(gs_pixel_loss_opts o, int64_t n,
                                                               const float* __restrict__ opacity,
                                                               const float* __restrict__ depth,
                                                               const float* __restrict__ sky,
                                                               const float* __restrict__ valid,
                                                               const float* __restrict__ lidar,
                                                               const uint8_t* __restrict__ select,
                                                               const float* __restrict__ fwd_out,
                                                               const float* __restrict__ g_out,
                                                               float* __restrict__ v_opacity, float* __restrict__ v_depth) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float vm = valid ? valid[i] : 1.f, op = opacity[i];
  const float inv_n = 1.f / (float)n;
  float vo = 0.f;
  if (sky) {
    const float p = op * vm, g = (1.f - sky[i]) * vm;
    if (o.bce_mode == 1) {
      // SafeBCE.backward (models/losses.py:56-77): x clipped away from the wrong end only, no gradient where x == y
      const float x0 = fminf(fmaxf(p, 0.f), 1.f), y = fminf(fmaxf(g, 0.f), 1.f);
      const float x = (y == 0.f) ? fminf(fmaxf(x0, 0.f), 1.f - o.bce_limit) : fminf(fmaxf(x0, o.bce_limit), 1.f);
      const float gx = (x == y) ? 0.f : ((y == 0.f) ? 1.f / (1.f - x) : -1.f / x);
      vo += g_out[0] * inv_n * vm * gx;
    } else {
      // torch's binary_cross_entropy backward: (p - g) / max(p (1 - p), 1e-12)
      vo += g_out[0] * inv_n * vm * (p - g) / fmaxf((1.f - p) * p, 1e-12f);
    }
  }
  if (op >= 1e-6f && op <= 1.f - 1e-6f) vo += g_out[3] * inv_n * -(logf(op) + 1.f);
  v_opacity[i] = vo;
  if (v_depth) {
    float vd = 0.f;
    if (lidar) {
      const DepthTerm d = depth_term(o, depth[i], lidar[i], vm, select, i);
      if (d.sel) vd = g_out[1] / (float)depth_denominator(o.depth_reduction, (double)fwd_out[2], n) * d.dloss_dpred;
    }
    v_depth[i] = vd;
  }
}

constexpr int PL_BLOCKS = 1024;

// ---- kornia.losses.inverse_depth_smoothness_loss on idepth = 1 / (depth + 1e-5) (base.py:575-585) ------------------
//   mean_{y, x < W-1} |id(y,x) - id(y,x+1)| exp(-mean_c |I(y,x,c) - I(y,x+1,c)|)  +  the same along y
// (kornia is fed the inverse depth repeated over 3 channels: the channel mean of identical values).
__device__ __forceinline__ float img_diff_weight(const float* __restrict__ img, int64_t a, int64_t b) {
  const float d = fabsf(img[3 * a] - img[3 * b]) + fabsf(img[3 * a + 1] - img[3 * b + 1]) +
                  fabsf(img[3 * a + 2] - img[3 * b + 2]);
  return __expf(-d * (1.f / 3.f));
}

__global__ __launch_bounds__(256) void inv_depth_smooth_fwd_kernel(int H, int W, const float* __restrict__ depth,
                                                                   const float* __restrict__ img,
                                                                   float* __restrict__ partials) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t n = (int64_t)H * W;
  float sx = 0.f, sy = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W), y = (int)(i / W);
    const float id = 1.f / (depth[i] + 1e-5f);
    if (x + 1 < W) sx += fabsf(id - 1.f / (depth[i + 1] + 1e-5f)) * img_diff_weight(img, i, i + 1);
    if (y + 1 < H) sy += fabsf(id - 1.f / (depth[i + W] + 1e-5f)) * img_diff_weight(img, i, i + W);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sx += __shfl_down(sx, off); sy += __shfl_down(sy, off); }
  if (lane == 0) { red[0][wave] = sx; red[1][wave] = sy; }
  __syncthreads();
  if (tid < 2) partials[(int64_t)tid * gridDim.x + blockIdx.x] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

__global__ __launch_bounds__(256) void inv_depth_smooth_reduce_kernel(int nb, int H, int W,
                                                                      const float* __restrict__ partials,
                                                                      float* __restrict__ out) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[2] = {0.0, 0.0};
  for (int i = tid; i < nb; i += 256) { a[0] += (double)partials[i]; a[1] += (double)partials[(int64_t)nb + i]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    if (lane == 0) red[k][wave] = a[k];
  }
  __syncthreads();
  if (tid == 0) {
    const double sx = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double sy = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    // (a mean over an empty set -- W == 1 or H == 1 -- is NaN in torch as well)
    out[0] = (float)(sx / ((double)H * (W - 1)) + sy / ((double)(H - 1) * W));
  }
}

__global__ __launch_bounds__(256) void inv_depth_smooth_bwd_kernel(int H, int W, const float* __restrict__ depth,
                                                                   const float* __restrict__ img,
                                                                   const float* __restrict__ g_out,
                                                                   float* __restrict__ v_depth) {
  const int64_t n = (int64_t)H * W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W), y = (int)(i / W);
  const float dd = depth[i] + 1e-5f, id = 1.f / dd;
  const float cx = 1.f / ((float)H * (float)(W - 1)), cy = 1.f / ((float)(H - 1) * (float)W);
  auto sgn = [](float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); };
  float g = 0.f;  // d loss / d id(y, x)
  if (x + 1 < W) g += cx * sgn(id - 1.f / (depth[i + 1] + 1e-5f)) * img_diff_weight(img, i, i + 1);
  if (x > 0) g -= cx * sgn(1.f / (depth[i - 1] + 1e-5f) - id) * img_diff_weight(img, i - 1, i);
  if (y + 1 < H) g += cy * sgn(id - 1.f / (depth[i + W] + 1e-5f)) * img_diff_weight(img, i, i + W);
  if (y > 0) g -= cy * sgn(1.f / (depth[i - W] + 1e-5f) - id) * img_diff_weight(img, i - W, i);
  v_depth[i] = g_out[0] * g * (-id * id);
}

// ---- dynamic-region L1 (base.py:598-613): |gt - pred| over the pixels with region > thresh (and valid), 3 channels ----
__global__ __launch_bounds__(256) void region_l1_fwd_kernel(int64_t n, const float* __restrict__ pred,
                                                            const float* __restrict__ gt,
                                                            const float* __restrict__ valid,
                                                            const float* __restrict__ region, float thresh,
                                                            float* __restrict__ partials) {
  __shared__ float red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f, c = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += (int64_t)gridDim.x * 256) {
    const float vm = valid ? valid[i] : 1.f;
    if (region[i] > thresh && vm != 0.f) {
      s += fabsf(gt[3 * i] * vm - pred[3 * i] * vm) + fabsf(gt[3 * i + 1] * vm - pred[3 * i + 1] * vm) +
           fabsf(gt[3 * i + 2] * vm - pred[3 * i + 2] * vm);
      c += 1.f;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s += __shfl_down(s, off); c += __shfl_down(c, off); }
  if (lane == 0) { red[0][wave] = s; red[1][wave] = c; }
  __syncthreads();
  if (tid < 2) partials[(int64_t)tid * gridDim.x + blockIdx.x] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

__global__ __launch_bounds__(256) void region_l1_reduce_kernel(int nb, const float* __restrict__ partials,
                                                               float* __restrict__ out) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double a[2] = {0.0, 0.0};
  for (int i = tid; i < nb; i += 256) { a[0] += (double)partials[i]; a[1] += (double)partials[(int64_t)nb + i]; }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a[k] += __shfl_down(a[k], off);
    if (lane == 0) red[k][wave] = a[k];
  }
  __syncthreads();
  if (tid == 0) {
    const double s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const double c = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    out[0] = c > 0.0 ? (float)(s / (3.0 * c)) : 0.f;  // the reference adds no term when the region is empty (:608)
    out[1] = (float)c;
  }
}

__global__ __launch_bounds__(256) void region_l1_bwd_kernel(int64_t n, const float* __restrict__ pred,
                                                            const float* __restrict__ gt,
                                                            const float* __restrict__ valid,
                                                            const float* __restrict__ region, float thresh,
                                                            const float* __restrict__ fwd_out,
                                                            const float* __restrict__ g_out, float* __restrict__ v_pred) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float vm = valid ? valid[i] : 1.f;
  float v[3] = {0.f, 0.f, 0.f};
  if (region[i] > thresh && vm != 0.f && fwd_out[1] > 0.f) {
    const float k = g_out[0] * vm / (3.f * fwd_out[1]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
      const float d = pred[3 * i + ch] * vm - gt[3 * i + ch] * vm;
      v[ch] = k * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    }
  }
  v_pred[3 * i] = v[0]; v_pred[3 * i + 1] = v[1]; v_pred[3 * i + 2] = v[2];
}

// ---- affine regulariser (base.py:587-596): |A[:3,:3] - I|.mean() + |A[:3,3]|.mean() over m [rows x 4] matrices ----
__global__ __launch_bounds__(64) void affine_reg_kernel(int m, int rows, const float* __restrict__ A,
                                                        const float* __restrict__ g_out, float* __restrict__ out,
                                                        float* __restrict__ v_A) {
  // one wave: m is the number of images of a step (1 in the reference)
  const int lane = threadIdx.x;
  float s_rot = 0.f, s_shift = 0.f;
  for (int e = lane; e < m * 12; e += 64) {
    const int k = e / 12, r = (e % 12) / 4, c = e % 4;
    const float a = A[(int64_t)k * rows * 4 + r * 4 + c];
    const float d = a - ((c < 3 && r == c) ? 1.f : 0.f);
    if (c < 3) s_rot += fabsf(d); else s_shift += fabsf(d);
    if (v_A) {
      const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
      v_A[(int64_t)k * rows * 4 + r * 4 + c] = g_out[0] * sg / (c < 3 ? (float)(m * 9) : (float)(m * 3));
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s_rot += __shfl_down(s_rot, off); s_shift += __shfl_down(s_shift, off); }
  if (lane == 0 && out) out[0] = s_rot / (float)(m * 9) + s_shift / (float)(m * 3);
}

}  // namespace gs

static gs_pixel_loss_opts default_pixel_loss_opts(float max_depth) {
  gs_pixel_loss_opts o;
  o.bce_mode = 0; o.bce_limit = 0.1f; o.depth_type = 0; o.depth_normalize = 0; o.depth_inverse = 0;
  o.depth_reduction = 0; o.max_depth = max_depth;
  return o;
}

static int check_pixel_loss_opts(const gs_pixel_loss_opts* o) {
  if (!o) return GS_ERR_INVALID_ARG;
  if (o->bce_mode < 0 || o->bce_mode > 1 || o->depth_type < 0 || o->depth_type > 2 || o->depth_reduction < 0 ||
      o->depth_reduction > 2 || !(o->max_depth > 0.f))
    return GS_ERR_INVALID_ARG;
  if (o->bce_mode == 1 && !(o->bce_limit > 0.f && o->bce_limit < 1.f)) return GS_ERR_INVALID_ARG;
  return GS_OK;
}

extern "C" int64_t gs_pixel_losses_workspace_floats(void) { return 4 * gs::PL_BLOCKS; }

extern "C" int gs_pixel_losses_ex_fwd(int64_t n, const gs_pixel_loss_opts* opts, const float* opacity,
                                      const float* depth, const float* sky_mask, const float* valid_mask,
                                      const float* lidar_depth, const uint8_t* select, float* err_map, float* workspace,
                                      int64_t workspace_floats, float* out, void* stream) {
  int st = check_pixel_loss_opts(opts);
  if (st != GS_OK) return st;
  if (n < 1 || !opacity || !workspace || !out || (lidar_depth && !depth)) return GS_ERR_INVALID_ARG;
  if (workspace_floats < gs_pixel_losses_workspace_floats()) return GS_ERR_WORKSPACE;
  const int nb = (int)std::min<int64_t>(gs::PL_BLOCKS, gs::ceil_div(n, 256));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs::pixel_losses_fwd_kernel, dim3(nb), dim3(256), 0, s, *opts, n, opacity, depth, sky_mask,
                     valid_mask, lidar_depth, select, err_map, workspace);
  hipLaunchKernelGGL(gs::pixel_losses_reduce_kernel, dim3(1), dim3(256), 0, s, nb, n, opts->depth_reduction, workspace,
                     out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_pixel_losses_ex_bwd(int64_t n, const gs_pixel_loss_opts* opts, const float* opacity,
                                      const float* depth, const float* sky_mask, const float* valid_mask,
                                      const float* lidar_depth, const uint8_t* select, const float* fwd_out,
                                      const float* g_out, float* v_opacity, float* v_depth, void* stream) {
  int st = check_pixel_loss_opts(opts);
  if (st != GS_OK) return st;
  if (n < 1 || !opacity || !fwd_out || !g_out || !v_opacity || (lidar_depth && !depth)) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::pixel_losses_bwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, *opts, n, opacity, depth, sky_mask, valid_mask, lidar_depth, select, fwd_out,
                     g_out, v_opacity, v_depth);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// out [4] = {sky BCE mean, depth L1 mean over the selected pixels, their count, opacity-entropy mean} over n = H*W pixels
// in the shipped configuration (bce, l1, not normalised, mean_on_hit).  sky_mask / lidar_depth may be NULL (that term is
// 0 / NaN); depth may be NULL when lidar_depth is.
extern "C" int gs_pixel_losses_fwd(int64_t n, const float* opacity, const float* depth, const float* sky_mask,
                                   const float* valid_mask, const float* lidar_depth, float max_depth, float* workspace,
                                   int64_t workspace_floats, float* out, void* stream) {
  const gs_pixel_loss_opts o = default_pixel_loss_opts(max_depth);
  return gs_pixel_losses_ex_fwd(n, &o, opacity, depth, sky_mask, valid_mask, lidar_depth, nullptr, nullptr, workspace,
                                workspace_floats, out, stream);
}

// v_opacity [n] (and v_depth [n], may be NULL) from g_out [4] (upstream gradients of out; entry 2 is ignored) and the
// forward's out (for the selected-pixel count).
extern "C" int gs_pixel_losses_bwd(int64_t n, const float* opacity, const float* depth, const float* sky_mask,
                                   const float* valid_mask, const float* lidar_depth, float max_depth,
                                   const float* fwd_out, const float* g_out, float* v_opacity, float* v_depth,
                                   void* stream) {
  const gs_pixel_loss_opts o = default_pixel_loss_opts(max_depth);
  return gs_pixel_losses_ex_bwd(n, &o, opacity, depth, sky_mask, valid_mask, lidar_depth, nullptr, fwd_out, g_out,
                                v_opacity, v_depth, stream);
}

extern "C" int gs_inv_depth_smooth_fwd(int32_t H, int32_t W, const float* depth, const float* image, float* workspace,
                                       int64_t workspace_floats, float* out, void* stream) {
  if (H < 1 || W < 1 || !depth || !image || !workspace || !out) return GS_ERR_INVALID_ARG;
  if (workspace_floats < gs_pixel_losses_workspace_floats()) return GS_ERR_WORKSPACE;
  const int nb = (int)std::min<int64_t>(gs::PL_BLOCKS, gs::ceil_div((int64_t)H * W, 256));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs::inv_depth_smooth_fwd_kernel, dim3(nb), dim3(256), 0, s, H, W, depth, image, workspace);
  hipLaunchKernelGGL(gs::inv_depth_smooth_reduce_kernel, dim3(1), dim3(256), 0, s, nb, H, W, workspace, out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_inv_depth_smooth_bwd(int32_t H, int32_t W, const float* depth, const float* image,
                                       const float* g_out, float* v_depth, void* stream) {
  if (H < 1 || W < 1 || !depth || !image || !g_out || !v_depth) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::inv_depth_smooth_bwd_kernel, dim3((unsigned)gs::ceil_div((int64_t)H * W, 256)), dim3(256), 0,
                     (hipStream_t)stream, H, W, depth, image, g_out, v_depth);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_region_l1_fwd(int64_t n, const float* pred, const float* gt, const float* valid_mask,
                                const float* region, float thresh, float* workspace, int64_t workspace_floats,
                                float* out, void* stream) {
  if (n < 1 || !pred || !gt || !region || !workspace || !out) return GS_ERR_INVALID_ARG;
  if (workspace_floats < gs_pixel_losses_workspace_floats()) return GS_ERR_WORKSPACE;
  const int nb = (int)std::min<int64_t>(gs::PL_BLOCKS, gs::ceil_div(n, 256));
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gs::region_l1_fwd_kernel, dim3(nb), dim3(256), 0, s, n, pred, gt, valid_mask, region, thresh,
                     workspace);
  hipLaunchKernelGGL(gs::region_l1_reduce_kernel, dim3(1), dim3(256), 0, s, nb, workspace, out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_region_l1_bwd(int64_t n, const float* pred, const float* gt, const float* valid_mask,
                                const float* region, float thresh, const float* fwd_out, const float* g_out,
                                float* v_pred, void* stream) {
  if (n < 1 || !pred || !gt || !region || !fwd_out || !g_out || !v_pred) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::region_l1_bwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     n, pred, gt, valid_mask, region, thresh, fwd_out, g_out, v_pred);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_affine_reg(int32_t m, int32_t rows, const float* A, const float* g_out, float* out, float* v_A,
                             void* stream) {
  if (m < 1 || rows < 3 || !A || (!out && !v_A) || (v_A && !g_out)) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::affine_reg_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, m, rows, A, g_out, out, v_A);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
