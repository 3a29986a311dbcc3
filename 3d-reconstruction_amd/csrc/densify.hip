// Densification statistics of one training step, fused.
// The reference keeps three per-Gaussian accumulators and updates them after every backward with ~10 boolean-mask
// indexing passes (/root/reference/3D-Reconstruction-main/models/trainers/base.py:279-297 scales the screen-space
// gradient to pixel units, models/gaussians/vanilla.py:154-182 `after_train` accumulates):
//   xys_grad_norm += |grad_px|   and   vis_counts += 1        where radii > 0
//   max_2Dsize     = max(max_2Dsize, radii / max(W, H))       where radii > 0
// (the first call initialises xys_grad_norm with |grad_px| for every row and vis_counts with ones).
// Here: one thread per Gaussian, one pass, reading radii [C,N] and the 2-float gradient straight out of the gradient
// record (row stride `ld` floats, so `means2d.absgrad` / `.grad` need no clone).  With C > 1 the per-camera terms add
// up (counts per visible camera, max over cameras) -- the semantics camera-sharded ranks then all-reduce.
#include "gs_common.hpp"

namespace gs {

__global__ __launch_bounds__(256) void densify_accumulate_kernel(int C, int N, const int32_t* __restrict__ radii,
                                                                 const float* __restrict__ grad2d, int64_t ld,
                                                                 float sx, float sy, float inv_last_size, int first,
                                                                 float* __restrict__ grad_norm_sum,
                                                                 float* __restrict__ vis_counts,
                                                                 float* __restrict__ max_2dsize) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float gsum = 0.f, gall = 0.f, cnt = 0.f, rmax = 0.f;
  for (int c = 0; c < C; ++c) {
    const int64_t idx = (int64_t)c * N + n;
    const int r = radii[idx];
    if (r > 0 || first) {
      const float gx = grad2d[idx * ld] * sx, gy = grad2d[idx * ld + 1] * sy;
      const float g = sqrtf(gx * gx + gy * gy);
      gall += g;
      if (r > 0) {
        gsum += g;
        cnt += 1.f;
        rmax = fmaxf(rmax, (float)r * inv_last_size);
      }
    }
  }
  if (first) {
    grad_norm_sum[n] = gall;
    vis_counts[n] = 1.f;
    max_2dsize[n] = rmax;  // from zeros: max(0, r / last_size) on the visible rows
  } else if (cnt > 0.f) {
    grad_norm_sum[n] += gsum;
    vis_counts[n] += cnt;
    max_2dsize[n] = fmaxf(max_2dsize[n], rmax);
  }
}

}  // namespace gs

extern "C" int gs_densify_accumulate(int32_t C, int32_t N, const int32_t* radii, const float* grad2d, int64_t ld,
                                     float scale_x, float scale_y, float inv_last_size, int32_t first,
                                     float* grad_norm_sum, float* vis_counts, float* max_2dsize, void* stream) {
  if (C < 1 || N < 0 || ld < 2) return GS_ERR_INVALID_ARG;
  if (N == 0) return GS_OK;
  if (!radii || !grad2d || !grad_norm_sum || !vis_counts || !max_2dsize) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::densify_accumulate_kernel, dim3((unsigned)gs::ceil_div(N, 256)), dim3(256), 0,
                     (hipStream_t)stream, C, N, radii, grad2d, ld, scale_x, scale_y, inv_last_size, first,
                     grad_norm_sum, vis_counts, max_2dsize);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
