// Densification / culling of one Gaussian class on the device (second half).
// The reference's `refinement_after` (/root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:197-283) grows
// and prunes the six Parameter tensors and their Adam moments with ~60 torch ops (boolean-mask gathers, `repeat`,
// `cat`, `bmm`, six `.item()` / print synchronisations) in two rounds -- split + duplicate (:208-259 with
// split_gaussians :314-342, dup_gaussians :344-357, basics.py:183-206 dup_in_optim), then cull on the grown arrays
// (:262-265, cull_gaussians :285-312, basics.py:162-181 remove_from_optim).  Here the two rounds are composed:
//   gs_refine_decide  one thread per old row: split / dup flags and which of {the row itself, its split children, its
//                     duplicate} survive the cull -> four 0/1 count columns
//   (gs_exclusive_scan_i32 over the four columns)
//   gs_refine_plan    one thread per old row: writes (source row, kind, sample index) for each of its output rows, in
//                     the reference's order [survivors | split children sample-major | duplicates]
//   gs_refine_gather_rows / gs_refine_split_means   build each new tensor in one pass from the plan
// so every tensor is read once and written once, and the host synchronises once (to size the outputs and the
// `randn` draw).  HBM-bound row gathers.
#include "gs_common.hpp"

namespace gs {

enum { RF_SPLIT = 1, RF_DUP = 2, RF_KEEP_SELF = 4, RF_KEEP_SPLIT = 8, RF_KEEP_DUP = 16 };
enum { KIND_OLD = 0, KIND_DUP = 1, KIND_SPLIT0 = 2 };  // kind >= 2: split child, sample kind - 2
enum { SCALE_3D = 0, SCALE_BALL = 1, SCALE_2D = 2 };

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }
// torch divides by a host scalar as a multiplication by its fp32 reciprocal (vanilla.py:338-339, size_fac = 1.6)
__device__ __forceinline__ float shrink(float raw) { return logf(expf(raw) * (1.f / 1.6f)); }

// max over the activated scale axes: `get_scaling.max(dim=-1)` for the densify tests (vanilla.py:112-128, 2-D Gaussians
// have a zero third axis), `exp(_scales).max(dim=-1)` over the raw axes for the cull test (vanilla.py:295)
__device__ __forceinline__ float max_scale(const float* __restrict__ raw, int64_t i, int mode, bool shrunk, bool raw_axes) {
  if (mode == SCALE_BALL) return expf(shrunk ? shrink(raw[i]) : raw[i]);
  float a = raw[3 * i], b = raw[3 * i + 1], c = raw[3 * i + 2];
  if (shrunk) { a = shrink(a); b = shrink(b); c = shrink(c); }
  const float m = fmaxf(expf(a), expf(b));
  return (mode == SCALE_2D && !raw_axes) ? fmaxf(m, 0.f) : fmaxf(m, expf(c));
}

struct RefineCfg {
  float densify_grad_thresh, densify_size_thresh;  // size threshold already multiplied by scene_scale
  float split_screen_size;                         // < 0: not applied (step >= stop_screen_size_at)
  float cull_alpha_thresh;
  float cull_scale_thresh;                         // already x scene_scale; < 0: not applied (step <= reset_alpha_interval)
  float cull_screen_size;                          // < 0: not applied
  int32_t do_densify, do_cull, scale_mode;
};

__global__ __launch_bounds__(256) void refine_decide_kernel(int64_t n, RefineCfg cfg, const float* __restrict__ raw_scales,
                                                            const float* __restrict__ raw_opac,
                                                            const float* __restrict__ grad_norm,
                                                            const float* __restrict__ vis_counts,
                                                            const float* __restrict__ max2d, uint8_t* __restrict__ flags,
                                                            int32_t* __restrict__ counts /* [4n+1] */) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) counts[4 * n] = 0;
  if (i >= n) return;
  uint32_t f = 0;
  if (cfg.do_densify) {
    const bool high = grad_norm[i] / vis_counts[i] > cfg.densify_grad_thresh;
    bool split = max_scale(raw_scales, i, cfg.scale_mode, false, false) > cfg.densify_size_thresh;
    if (cfg.split_screen_size >= 0.f) split |= max2d[i] > cfg.split_screen_size;
    split &= high;
    // the duplicate test runs after split_gaussians has shrunk the split rows in place (vanilla.py:339 then :229-233)
    const bool dup = high && max_scale(raw_scales, i, cfg.scale_mode, split, false) <= cfg.densify_size_thresh;
    f |= (split ? RF_SPLIT : 0) | (dup ? RF_DUP : 0);
  }
  bool cull_base = false, cull_screen = false;
  if (cfg.do_cull) {
    cull_base = sigmoidf(raw_opac[i]) < cfg.cull_alpha_thresh;
    if (cfg.cull_scale_thresh >= 0.f) {
      cull_base |= max_scale(raw_scales, i, cfg.scale_mode, (f & RF_SPLIT) != 0, true) > cfg.cull_scale_thresh;
      // new rows enter with max_2Dsize = 0 (vanilla.py:250-253): only the old row can fail the screen-size test
      if (cfg.cull_screen_size >= 0.f) cull_screen = max2d[i] > cfg.cull_screen_size;
    }
  }
  if (!(cull_base || cull_screen)) f |= RF_KEEP_SELF;
  if ((f & RF_SPLIT) && !cull_base) f |= RF_KEEP_SPLIT;
  if ((f & RF_DUP) && !cull_base) f |= RF_KEEP_DUP;
  flags[i] = (uint8_t)f;
  counts[i] = (f & RF_SPLIT) ? 1 : 0;
  counts[n + i] = (f & RF_KEEP_SELF) ? 1 : 0;
  counts[2 * n + i] = (f & RF_KEEP_SPLIT) ? 1 : 0;
  counts[3 * n + i] = (f & RF_KEEP_DUP) ? 1 : 0;
}

// scan = exclusive scan of counts [4n+1]; column bases scan[k*n], column totals scan[(k+1)*n] - scan[k*n]
__global__ __launch_bounds__(256) void refine_plan_kernel(int64_t n, int nsamps, const uint8_t* __restrict__ flags,
                                                          const int32_t* __restrict__ scan, int32_t* __restrict__ src,
                                                          uint8_t* __restrict__ kind, int32_t* __restrict__ sample_idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t f = flags[i];
  const int32_t n_splits = scan[n] - scan[0];
  const int32_t n_self = scan[2 * n] - scan[n];
  const int32_t n_ksplit = scan[3 * n] - scan[2 * n];
  if (f & RF_KEEP_SELF) {
    const int32_t r = scan[n + i] - scan[n];
    src[r] = (int32_t)i; kind[r] = KIND_OLD; sample_idx[r] = -1;
  }
  if (f & RF_KEEP_SPLIT) {
    const int32_t rank_all = scan[i] - scan[0];            // among all split rows: indexes the randn draw
    const int32_t rank_kept = scan[2 * n + i] - scan[2 * n];
    for (int s = 0; s < nsamps; ++s) {
      const int32_t r = n_self + s * n_ksplit + rank_kept;
      src[r] = (int32_t)i; kind[r] = (uint8_t)(KIND_SPLIT0 + s); sample_idx[r] = s * n_splits + rank_all;
    }
  }
  if (f & RF_KEEP_DUP) {
    const int32_t r = n_self + nsamps * n_ksplit + (scan[3 * n + i] - scan[3 * n]);
    src[r] = (int32_t)i; kind[r] = KIND_DUP; sample_idx[r] = -1;
  }
}

enum { GATHER_COPY = 0, GATHER_ZERO_NEW = 1, GATHER_SCALES = 2 };

// out [n_out, W] <- in [n, W] rows by the plan; a workgroup builds 256 consecutive output rows as one contiguous block
// (coalesced writes, row-contiguous gathered reads)
__global__ __launch_bounds__(256) void refine_gather_kernel(int64_t n_out, int W, float inv_w, int mode,
                                                            const int32_t* __restrict__ src,
                                                            const uint8_t* __restrict__ kind,
                                                            const uint8_t* __restrict__ flags,
                                                            const float* __restrict__ in, float* __restrict__ out) {
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int rows = (int)min((int64_t)256, n_out - row0);
  float* base = out + row0 * W;
  for (int m = threadIdx.x; m < rows * W; m += 256) {
    int r = (int)(((float)m + 0.5f) * inv_w);  // m < 256 * W <= 2^16: exact up to the +-1 fixed below
    int k = m - r * W;
    if (k < 0) { --r; k += W; } else if (k >= W) { ++r; k -= W; }
    const int32_t s = src[row0 + r];
    float v;
    if (mode == GATHER_ZERO_NEW && kind[row0 + r] != KIND_OLD) {
      v = 0.f;  // fresh Adam moments for split children and duplicates (basics.py:191-202)
    } else {
      v = in[(int64_t)s * W + k];
      if (mode == GATHER_SCALES && (flags[s] & RF_SPLIT)) v = shrink(v);  // parent, children and its duplicate alike
    }
    base[m] = v;
  }
}

// new means: split children get mean + R(q / |q|) (exp(scale) * sample) (vanilla.py:321-329), everything else a copy
__global__ __launch_bounds__(256) void refine_means_kernel(int64_t n_out, int scale_mode, const int32_t* __restrict__ src,
                                                           const uint8_t* __restrict__ kind,
                                                           const int32_t* __restrict__ sample_idx,
                                                           const float* __restrict__ means,
                                                           const float* __restrict__ raw_scales,
                                                           const float* __restrict__ raw_quats,
                                                           const float* __restrict__ samples, float* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_out) return;
  const int64_t s = src[r];
  float mx = means[3 * s], my = means[3 * s + 1], mz = means[3 * s + 2];
  if (kind[r] >= KIND_SPLIT0) {
    const int64_t j = sample_idx[r];
    float sx, sy, sz;
    if (scale_mode == SCALE_BALL) {
      sx = sy = sz = expf(raw_scales[s]);
    } else {
      sx = expf(raw_scales[3 * s]);
      sy = expf(raw_scales[3 * s + 1]);
      sz = scale_mode == SCALE_2D ? 0.f : expf(raw_scales[3 * s + 2]);
    }
    const float px = sx * samples[3 * j], py = sy * samples[3 * j + 1], pz = sz * samples[3 * j + 2];
    const float4 q = reinterpret_cast<const float4*>(raw_quats)[s];
    const float qn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    const float w = q.x / qn, x = q.y / qn, y = q.z / qn, z = q.w / qn;
    // quat_to_rotmat (wxyz), R/models/gaussians/basics.py:14
    const float r00 = 1.f - 2.f * (y * y + z * z), r01 = 2.f * (x * y - w * z), r02 = 2.f * (x * z + w * y);
    const float r10 = 2.f * (x * y + w * z), r11 = 1.f - 2.f * (x * x + z * z), r12 = 2.f * (y * z - w * x);
    const float r20 = 2.f * (x * z - w * y), r21 = 2.f * (y * z + w * x), r22 = 1.f - 2.f * (x * x + y * y);
    mx += r00 * px + r01 * py + r02 * pz;
    my += r10 * px + r11 * py + r12 * pz;
    mz += r20 * px + r21 * py + r22 * pz;
  }
  out[3 * r] = mx;
  out[3 * r + 1] = my;
  out[3 * r + 2] = mz;
}

}  // namespace gs

extern "C" int gs_refine_decide(int64_t n, int32_t scale_mode, int32_t do_densify, int32_t do_cull,
                                float densify_grad_thresh, float densify_size_thresh, float split_screen_size,
                                float cull_alpha_thresh, float cull_scale_thresh, float cull_screen_size,
                                const float* raw_scales, const float* raw_opacities, const float* grad_norm_sum,
                                const float* vis_counts, const float* max_2dsize, uint8_t* flags, int32_t* counts,
                                void* stream) {
  if (n < 0 || n > (int64_t)500000000 || scale_mode < 0 || scale_mode > 2) return GS_ERR_INVALID_ARG;
  if (!counts) return GS_ERR_INVALID_ARG;
  if (n > 0 && (!raw_scales || !raw_opacities || !flags)) return GS_ERR_INVALID_ARG;
  if (n > 0 && do_densify && (!grad_norm_sum || !vis_counts)) return GS_ERR_INVALID_ARG;
  const bool needs_max2d = (do_densify && split_screen_size >= 0.f) ||
                           (do_cull && cull_scale_thresh >= 0.f && cull_screen_size >= 0.f);
  if (n > 0 && needs_max2d && !max_2dsize) return GS_ERR_INVALID_ARG;
  gs::RefineCfg cfg{densify_grad_thresh, densify_size_thresh, split_screen_size, cull_alpha_thresh, cull_scale_thresh,
                    cull_screen_size, do_densify, do_cull, scale_mode};
  hipLaunchKernelGGL(gs::refine_decide_kernel, dim3((unsigned)gs::ceil_div(n > 0 ? n : 1, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, cfg, raw_scales, raw_opacities, grad_norm_sum, vis_counts, max_2dsize,
                     flags, counts);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_refine_plan(int64_t n, int32_t n_split_samples, const uint8_t* flags, const int32_t* scan,
                              int32_t* src_rows, uint8_t* kinds, int32_t* sample_idx, void* stream) {
  if (n < 0 || n_split_samples < 0 || n_split_samples > 250) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!flags || !scan || !src_rows || !kinds || !sample_idx) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::refine_plan_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                     n_split_samples, flags, scan, src_rows, kinds, sample_idx);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_refine_gather_rows(int64_t n_out, int32_t W, int32_t mode, const int32_t* src_rows,
                                     const uint8_t* kinds, const uint8_t* flags, const float* in, float* out,
                                     void* stream) {
  if (n_out < 0 || W < 0 || W > 256 || mode < 0 || mode > 2) return GS_ERR_INVALID_ARG;
  if (n_out == 0 || W == 0) return GS_OK;
  if (!src_rows || !kinds || !in || !out || (mode == gs::GATHER_SCALES && !flags)) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::refine_gather_kernel, dim3((unsigned)gs::ceil_div(n_out, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_out, W, 1.f / (float)W, mode, src_rows, kinds, flags, in, out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

extern "C" int gs_refine_split_means(int64_t n_out, int32_t scale_mode, const int32_t* src_rows, const uint8_t* kinds,
                                     const int32_t* sample_idx, const float* means, const float* raw_scales,
                                     const float* raw_quats, const float* samples, float* out_means, void* stream) {
  if (n_out < 0 || scale_mode < 0 || scale_mode > 2) return GS_ERR_INVALID_ARG;
  if (n_out == 0) return GS_OK;
  if (!src_rows || !kinds || !sample_idx || !means || !raw_scales || !raw_quats || !out_means) return GS_ERR_INVALID_ARG;
  if ((uintptr_t)raw_quats & 15) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::refine_means_kernel, dim3((unsigned)gs::ceil_div(n_out, 256)), dim3(256), 0,
                     (hipStream_t)stream, n_out, scale_mode, src_rows, kinds, sample_idx, means, raw_scales, raw_quats,
                     samples, out_means);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// ---- opacity reset (VanillaGaussians.refinement_after, R/models/gaussians/vanilla.py:268-280) ----------------------
//   raw = logit(min(sigmoid(raw), reset_alpha_value));  Adam exp_avg = exp_avg_sq = 0 for the opacity parameter
// The reference does it with four elementwise torch ops and two zeros_like allocations; here one pass, in place.
namespace gs {
__global__ __launch_bounds__(256) void opacity_reset_kernel(int64_t n, float reset_value, float* __restrict__ raw,
                                                            float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float o = 1.f / (1.f + expf(-raw[i]));
  const float r = fminf(o, reset_value);
  raw[i] = logf(r / (1.f - r));  // torch.logit (no eps): +-inf at exactly 0 / 1, as the reference
  if (exp_avg) exp_avg[i] = 0.f;
  if (exp_avg_sq) exp_avg_sq[i] = 0.f;
}
}  // namespace gs

extern "C" int gs_opacity_reset(int64_t n, float reset_alpha_value, float* raw_opacities, float* exp_avg,
                                float* exp_avg_sq, void* stream) {
  if (n < 0 || !(reset_alpha_value > 0.f)) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!raw_opacities) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::opacity_reset_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, n,
                     reset_alpha_value, raw_opacities, exp_avg, exp_avg_sq);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
