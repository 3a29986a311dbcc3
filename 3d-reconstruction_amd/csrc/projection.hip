// fully_fused_projection forward/backward for gfx950, fused with:
//   * SH colour evaluation per (camera, Gaussian)      (gsplat rasterization(sh_degree=...) path)
//   * opacity x anti-aliasing compensation
//   * the depth channel of render_mode "RGB+D/ED"/"D"/"ED"
//   * the per-Gaussian tile bounding box + tile count of isect_tiles pass 1
// Semantics: SURVEY.md Appendix A.1 (gsplat fully_fused_projection / persp_proj / add_blur and their
// vjps), as reached from /root/reference/3D-Reconstruction-main/models/trainers/base.py:393-408.
//
// Work decomposition (MI355X-first): ONE THREAD PER GAUSSIAN, looping over the C cameras.
//   - the 40 B of Gaussian parameters (and the 12K B of SH coefficients) are read from HBM once, not
//     once per camera; the 3D covariance is built once;
//   - camera matrices are wave-uniform -> the compiler keeps them in SGPRs (s_load), no VGPR/LDS cost;
//   - the backward accumulates v_means/v_quats/v_scales/v_opacity/v_sh over cameras IN REGISTERS, so the
//     parameter gradients need no atomics at all (gsplat needs warp-reduce + atomics here);
//   - outputs are [C,N]-major, so every store instruction of a wave is contiguous across lanes
//     (radii/tiles 256 B, means2d 512 B, splat record 64 B per lane = one full HBM request each).
// HBM-bound: algorithmic bytes/Gaussian fwd = 40 (+12K SH when visible) + C*(4+4+8) + V/N*64.
#include "sh_math.hpp"

namespace gs {

struct Cam {
  float R[9];
  float t[3];
  float fx, fy, cx, cy;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ viewmats, const float* __restrict__ Ks, int c) {
  Cam cam;
  const float* vm = viewmats + c * 16;
  const float* K = Ks + c * 9;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) cam.R[i * 3 + j] = vm[i * 4 + j];
    cam.t[i] = vm[i * 4 + 3];
  }
  cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
  return cam;
}

// Camera centre -inv(R) t of a world->camera matrix (adjugate form, so R need not be orthonormal); used for the SH
// view directions when the caller passes no `campos` (wave-uniform: a handful of scalar ops per camera).
__device__ __forceinline__ void cam_center(const float* __restrict__ campos, const Cam& cam, int c, float cp[3]) {
  if (campos) {
    cp[0] = campos[3 * c]; cp[1] = campos[3 * c + 1]; cp[2] = campos[3 * c + 2];
    return;
  }
  const float* R = cam.R;
  const float c12[3] = {R[4] * R[8] - R[5] * R[7], R[5] * R[6] - R[3] * R[8], R[3] * R[7] - R[4] * R[6]};
  const float c20[3] = {R[7] * R[2] - R[8] * R[1], R[8] * R[0] - R[6] * R[2], R[6] * R[1] - R[7] * R[0]};
  const float c01[3] = {R[1] * R[5] - R[2] * R[4], R[2] * R[3] - R[0] * R[5], R[0] * R[4] - R[1] * R[3]};
  const float idet = 1.f / (R[0] * c12[0] + R[1] * c12[1] + R[2] * c12[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) cp[i] = -(cam.t[0] * c12[i] + cam.t[1] * c20[i] + cam.t[2] * c01[i]) * idet;
}

// wxyz (un-normalised) -> rotation matrix of the normalised quaternion; returns 1/|q| and the unit quaternion.
__device__ __forceinline__ void quat_to_rot(const float q[4], float Rq[9], float qn[4], float& inv_norm) {
  inv_norm = rsqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float w = q[0] * inv_norm, x = q[1] * inv_norm, y = q[2] * inv_norm, z = q[3] * inv_norm;
  qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
  Rq[0] = 1.f - 2.f * (y * y + z * z); Rq[1] = 2.f * (x * y - w * z);       Rq[2] = 2.f * (x * z + w * y);
  Rq[3] = 2.f * (x * y + w * z);       Rq[4] = 1.f - 2.f * (x * x + z * z); Rq[5] = 2.f * (y * z - w * x);
  Rq[6] = 2.f * (x * z - w * y);       Rq[7] = 2.f * (y * z + w * x);       Rq[8] = 1.f - 2.f * (x * x + y * y);
}

// Sigma = (Rq diag(s)) (Rq diag(s))^T, symmetric, stored {00,01,02,11,12,22}
__device__ __forceinline__ void covar3d(const float Rq[9], const float s[3], float S[6]) {
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * s[j];
  S[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  S[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  S[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  S[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  S[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  S[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// a0 b0 + a1 b1 (+ a2 b2).  PIN: with the fusion spelled out -- the forward kernel exists in two instantiations that must
// give one result (see projection_fwd_kernel); the backward, a single kernel, leaves it to the optimiser (0.002 ms).
template <bool PIN>
__device__ __forceinline__ float dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
  if constexpr (PIN) return fmaf(a2, b2, fmaf(a1, b1, a0 * b0));
  return a0 * b0 + a1 * b1 + a2 * b2;
}
template <bool PIN>
__device__ __forceinline__ float dot2(float a0, float b0, float a1, float b1) {
  if constexpr (PIN) return fmaf(a1, b1, a0 * b0);
  return a0 * b0 + a1 * b1;
}

__device__ __forceinline__ float sym(const float S[6], int i, int j) {
  // index into {00,01,02,11,12,22}
  constexpr int map[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
  return S[map[i * 3 + j]];
}

// Cc = R S R^T (symmetric 6)
template <bool PIN = false>
__device__ __forceinline__ void covar_world_to_cam(const float R[9], const float S[6], float Cc[6]) {
  float A[9];  // A = R S
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      A[i * 3 + j] = dot3<PIN>(R[i * 3 + 0], sym(S, 0, j), R[i * 3 + 1], sym(S, 1, j), R[i * 3 + 2], sym(S, 2, j));
  auto dot = [&](int i, int j) { return dot3<PIN>(A[i * 3 + 0], R[j * 3 + 0], A[i * 3 + 1], R[j * 3 + 1], A[i * 3 + 2], R[j * 3 + 2]); };
  Cc[0] = dot(0, 0); Cc[1] = dot(0, 1); Cc[2] = dot(0, 2);
  Cc[3] = dot(1, 1); Cc[4] = dot(1, 2); Cc[5] = dot(2, 2);
}

struct Proj2D {
  float mx, my;             // mean2d
  float c00, c01, c11;      // blurred 2D covariance
  float det;                // of the blurred covariance
  float comp;               // compensation
  float Ja, Jb, Jc, Jd;     // J = [[Ja,0,Jb],[0,Jc,Jd]]
  float tx, ty, rz;
  bool in_x, in_y;          // clamp gates
};

template <bool PIN = false>
__device__ __forceinline__ Proj2D persp_proj(const Cam& cam, const float mc[3], const float Cc[6], float W, float H,
                                             float eps2d) {
  Proj2D o;
  const float x = mc[0], y = mc[1], z = mc[2];
  const float tan_fovx = 0.5f * W / cam.fx, tan_fovy = 0.5f * H / cam.fy;
  const float lim_x_pos = (W - cam.cx) / cam.fx + 0.3f * tan_fovx;
  const float lim_x_neg = cam.cx / cam.fx + 0.3f * tan_fovx;
  const float lim_y_pos = (H - cam.cy) / cam.fy + 0.3f * tan_fovy;
  const float lim_y_neg = cam.cy / cam.fy + 0.3f * tan_fovy;
  const float rz = 1.f / z, rz2 = rz * rz;
  const float xr = x * rz, yr = y * rz;
  o.in_x = (xr <= lim_x_pos) && (xr >= -lim_x_neg);
  o.in_y = (yr <= lim_y_pos) && (yr >= -lim_y_neg);
  o.tx = z * fminf(lim_x_pos, fmaxf(-lim_x_neg, xr));
  o.ty = z * fminf(lim_y_pos, fmaxf(-lim_y_neg, yr));
  o.rz = rz;
  o.Ja = cam.fx * rz; o.Jb = -cam.fx * o.tx * rz2;
  o.Jc = cam.fy * rz; o.Jd = -cam.fy * o.ty * rz2;
  // rows of J*Cc
  const float v00 = dot2<PIN>(o.Ja, Cc[0], o.Jb, Cc[2]);
  const float v01 = dot2<PIN>(o.Ja, Cc[1], o.Jb, Cc[4]);
  const float v02 = dot2<PIN>(o.Ja, Cc[2], o.Jb, Cc[5]);
  const float v11 = dot2<PIN>(o.Jc, Cc[3], o.Jd, Cc[4]);
  const float v12 = dot2<PIN>(o.Jc, Cc[4], o.Jd, Cc[5]);
  const float c00 = dot2<PIN>(v00, o.Ja, v02, o.Jb);
  const float c01 = dot2<PIN>(v01, o.Jc, v02, o.Jd);
  const float c11 = dot2<PIN>(v11, o.Jc, v12, o.Jd);
  o.mx = PIN ? fmaf(cam.fx * x, rz, cam.cx) : cam.fx * x * rz + cam.cx;
  o.my = PIN ? fmaf(cam.fy * y, rz, cam.cy) : cam.fy * y * rz + cam.cy;
  const float det_orig = PIN ? fmaf(c00, c11, -(c01 * c01)) : c00 * c11 - c01 * c01;
  o.c00 = c00 + eps2d;
  o.c11 = c11 + eps2d;
  o.c01 = c01;
  o.det = PIN ? fmaf(o.c00, o.c11, -(c01 * c01)) : o.c00 * o.c11 - c01 * c01;
  o.comp = sqrtf(fmaxf(0.f, det_orig / o.det));
  return o;
}

// ------------------------------------------------------------------------------------------ forward
// PREFETCH (SH colours): the Gaussian's coefficient row is requested up front, behind the geometry loads, whether or not a
// camera will see it -- right when most Gaussians are seen by some camera (5-camera rig: 87 %).  !PREFETCH: the row is
// requested at the first camera that sees the Gaussian; with one or two cameras, or a few neighbouring views of a
// trajectory, three quarters of the rows are never needed (1 M Gaussians, one camera: 0.086 -> 0.048 ms; 6 M x 3
// trajectory views: 2177 -> 2416 views/s; five rig cameras: 0.111 -> 0.127 ms, which is why the host chooses).
// Two instantiations must give a camera the same bits whether it is projected alone or in a batch of five: the geometry
// helpers above therefore spell their multiply-adds out (fmaf) instead of leaving the fusion of a*b + c*d + e*f to the
// optimiser, which resolved it differently in the two kernels.
template <int CMODE, int DEG, bool PREFETCH = true>
__global__ __launch_bounds__(256, (CMODE == GS_COLOR_SH && DEG == 4) ? 1 : 4) void projection_fwd_kernel(
    gs_proj_params p, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ viewmats,
    const float* __restrict__ Ks, const float* __restrict__ campos, const float* __restrict__ colors,
    int32_t* __restrict__ radii, float* __restrict__ means2d, float* __restrict__ splats,
    int32_t* __restrict__ tiles_per_gauss, const float* __restrict__ covars, uint64_t* __restrict__ vis_words,
    int32_t* __restrict__ vis_counts) {
  // whole waves stay alive (cooperative loads below); lanes past N are clamped and never store
  __shared__ float s_tile[4][192];
  const int n_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const int64_t row0 = n_raw - lane;
  const int rows = (int)min((int64_t)64, (int64_t)p.N - row0);  // wave-uniform
  if (rows <= 0) return;
  const bool live = n_raw < p.N;
  const int n = live ? n_raw : p.N - 1;
  const int n_words = (p.N + 63) >> 6;
  float* tile = s_tile[threadIdx.x >> 6];
  float mu[3];
  const bool coop_ok = true;
  load3(means, coop_ok && aligned16(means), n, row0, rows, lane, tile, mu);
  // SH coefficients: the whole row up front (aligned float4 loads in flight behind the geometry), not inside the
  // per-camera branch where every visible camera would wait for them again
  constexpr int NB3 = (CMODE == GS_COLOR_SH) ? 3 * (DEG + 1) * (DEG + 1) : 1;
  float creg[NB3];
  bool have_row = false;
  if constexpr (CMODE == GS_COLOR_SH && PREFETCH) {
    sh_load_row<NB3>(colors + (int64_t)n * p.sh_K * 3, (p.sh_K % 4 == 0) && aligned16(colors), creg);
    have_row = true;
  }
  const float opac = opacities[n];
  float S[6];
  if (covars) {  // precomputed world covariances (gsplat `covars=`), upper triangle {xx,xy,xz,yy,yz,zz}
#pragma unroll
    for (int i = 0; i < 6; ++i) S[i] = covars[6 * n + i];
  } else {
    float q[4];
    if (aligned16(quats)) {
      const float4 q4 = reinterpret_cast<const float4*>(quats)[n];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = quats[4 * n + i];
    }
    float sc[3];
    load3(scales, coop_ok && aligned16(scales), n, row0, rows, lane, tile, sc);
    float Rq[9], qn[4], inv_norm;
    quat_to_rot(q, Rq, qn, inv_norm);
    covar3d(Rq, sc, S);
  }
  const float W = (float)p.width, H = (float)p.height;
  const float ts = (float)p.tile_size;

  for (int c = 0; c < p.C; ++c) {
    const int64_t idx = (int64_t)c * p.N + n;
    const Cam cam = load_cam(viewmats, Ks, c);
    float mc[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) mc[i] = fmaf(cam.R[i * 3 + 2], mu[2], fmaf(cam.R[i * 3 + 1], mu[1], fmaf(cam.R[i * 3], mu[0], cam.t[i])));
    bool ok = live && (mc[2] >= p.near_plane) && (mc[2] <= p.far_plane);
    Proj2D pr;
    float radius = 0.f;
    if (ok) {
      float Cc[6];
      covar_world_to_cam<true>(cam.R, S, Cc);
      pr = persp_proj<true>(cam, mc, Cc, W, H, p.eps2d);
      ok = pr.det > 0.f;
      if (ok) {
        const float b = 0.5f * (pr.c00 + pr.c11);
        const float v1 = b + sqrtf(fmaxf(0.01f, b * b - pr.det));
        radius = ceilf(3.f * sqrtf(v1));
        ok = radius > p.radius_clip;
        ok = ok && !(pr.mx + radius <= 0.f || pr.mx - radius >= W || pr.my + radius <= 0.f || pr.my - radius >= H);
      }
    }
    if (vis_words) {
      // visibility bitmap of this wave's 64 Gaussians for camera c + its population: what the packed colour-gradient
      // message of the backward (gs_projection_bwd_split) and its consumers index by (every live lane gets here)
      const unsigned long long vb = __ballot(ok);
      if ((threadIdx.x & 63) == 0) {
        vis_words[(int64_t)c * n_words + (n >> 6)] = vb;
        vis_counts[(int64_t)c * n_words + (n >> 6)] = __popcll(vb);
      }
    }
    if (!ok) {
      if (live) {
        radii[idx] = 0;
        tiles_per_gauss[idx] = 0;
        if (means2d) reinterpret_cast<float2*>(means2d)[idx] = make_float2(0.f, 0.f);
      }
      continue;
    }
    // tile bounding box (isect_tiles pass 1, fp32 exactly as Appendix A.3)
    const float tr = radius / ts, tx = pr.mx / ts, ty = pr.my / ts;
    const int x0 = (int)fminf(fmaxf(floorf(tx - tr), 0.f), (float)p.tile_w);
    const int x1 = (int)fminf(fmaxf(ceilf(tx + tr), 0.f), (float)p.tile_w);
    const int y0 = (int)fminf(fmaxf(floorf(ty - tr), 0.f), (float)p.tile_h);
    const int y1 = (int)fminf(fmaxf(ceilf(ty + tr), 0.f), (float)p.tile_h);
    const float idet = 1.f / pr.det;
    const float comp = p.calc_compensations ? pr.comp : 1.f;

    float ch[4] = {0.f, 0.f, 0.f, 0.f};
    int nc = 0;
    if constexpr (CMODE == GS_COLOR_N) {
      nc = p.color_dim;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < nc) ch[k] = colors[(int64_t)n * nc + k];
    } else if constexpr (CMODE == GS_COLOR_CN) {
      nc = p.color_dim;
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < nc) ch[k] = colors[idx * nc + k];
    } else if constexpr (CMODE == GS_COLOR_SH) {
      nc = 3;
      float rgb[3];
      float cp[3];
      cam_center(campos, cam, c, cp);
      if constexpr (!PREFETCH) {
        if (!have_row) {
          sh_load_row<NB3>(colors + (int64_t)n * p.sh_K * 3, (p.sh_K % 4 == 0) && aligned16(colors), creg);
          have_row = true;
        }
      }
      sh_eval<DEG>(creg, mu[0] - cp[0], mu[1] - cp[1], mu[2] - cp[2], rgb);
#pragma unroll
      for (int k = 0; k < 3; ++k) ch[k] = fmaxf(rgb[k] + 0.5f, 0.f);
    }
    if (p.depth_channel) {
      // nc <= 3 guaranteed by check_params
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k == nc) ch[k] = mc[2];
    }

    radii[idx] = (int32_t)radius;
    tiles_per_gauss[idx] = (y1 - y0) * (x1 - x0);
    if (means2d) reinterpret_cast<float2*>(means2d)[idx] = make_float2(pr.mx, pr.my);
    float4* rec = reinterpret_cast<float4*>(splats + idx * REC);
    rec[0] = make_float4(pr.mx, pr.my, pr.c11 * idet, -pr.c01 * idet);
    rec[1] = make_float4(pr.c00 * idet, opac * comp, ch[0], ch[1]);
    rec[2] = make_float4(ch[2], ch[3], mc[2], __int_as_float((int32_t)radius));
    rec[3] = make_float4(comp, __int_as_float(x0 | (y0 << 16)), __int_as_float(x1 | (y1 << 16)), 0.f);
  }
}

// ------------------------------------------------------------------------------------------ backward
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Layout of the packed colour-gradient message (gs_colour_message_layout): byte offsets of its sections.
struct MsgLayout {
  int64_t campos, words, prefix, vals, total;
};
__host__ __device__ inline MsgLayout msg_layout(int C, int N, int64_t cap) {
  const int64_t W = (N + 63) >> 6;
  auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
  MsgLayout L;
  L.campos = 16;  // behind the header {int32 n_cams, 12 B reserved}
  L.words = up(16 + (int64_t)C * 12);
  L.prefix = L.words + up((int64_t)C * W * 8);
  L.vals = L.prefix + up((int64_t)C * W * 4);
  L.total = L.vals + up(cap * 12);
  return L;
}

// SHSPLIT (only with CMODE == GS_COLOR_SH): the SH part of the backward is NOT done here.  Instead the clamp-gated
// colour gradient of every visible (camera, Gaussian) pair is written, packed in ascending (c, n) order, into the
// colour-gradient message together with the visibility words, their prefix and the camera centres; sh_bwd_packed_kernel
// turns any number of such messages (this rank's, or every rank's after an all-gather) into v_sh and the view-direction
// part of v_means.  Without the 48 SH accumulators the kernel needs a third of the registers.
template <int CMODE, int DEG, bool VIEWGRAD, bool SHSPLIT = false>
__global__ __launch_bounds__(256, SHSPLIT ? 4 : 2) void projection_bwd_kernel(
    gs_proj_params p, const float* __restrict__ means, const float* __restrict__ quats,
    const float* __restrict__ scales, const float* __restrict__ opacities, const float* __restrict__ viewmats,
    const float* __restrict__ Ks, const float* __restrict__ campos, const float* __restrict__ colors,
    const int32_t* __restrict__ radii, const float* __restrict__ splats, const float* __restrict__ v_means2d,
    int64_t ld_vm2, const float* __restrict__ v_splats, float* __restrict__ v_means, float* __restrict__ v_quats,
    float* __restrict__ v_scales, float* __restrict__ v_opacities, float* __restrict__ v_colors,
    float* __restrict__ v_viewmats, float* __restrict__ v_campos, const float* __restrict__ covars,
    float* __restrict__ v_covars, const int32_t* __restrict__ vis_prefix = nullptr,
    unsigned char* __restrict__ msg = nullptr, int64_t msg_cap = 0, int msg_cams = 0) {
  static_assert(!SHSPLIT || CMODE == GS_COLOR_SH, "SHSPLIT is the SH colour path");
  constexpr int NB = (CMODE == GS_COLOR_SH && !SHSPLIT) ? (DEG + 1) * (DEG + 1) : 1;
  const int n_raw = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = n_raw < p.N;
  const int n = live ? n_raw : p.N - 1;  // keep whole waves alive (VIEWGRAD reductions, cooperative loads / stores)
  __shared__ float s_tile3[4][192];
  float* tile3 = s_tile3[threadIdx.x >> 6];
  const int lane3 = threadIdx.x & 63;
  const int64_t row0_3 = n_raw - lane3;
  const int rows3 = (int)max((int64_t)0, min((int64_t)64, (int64_t)p.N - row0_3));  // wave-uniform
  // (measured: the cooperative [N,3] transfers that help the forward cost this kernel 0.004-0.006 ms)
  const bool coop3 = false;
  float mu[3];
  load3(means, aligned16(means) && coop3, n, row0_3, rows3, lane3, tile3, mu);
  const float opac = opacities[n];
  float Rq[9], qn[4], inv_norm = 0.f, S[6];
  float sc[3] = {0.f, 0.f, 0.f};
  if (covars) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S[i] = covars[6 * n + i];
  } else {
    float q[4];
    if (aligned16(quats)) {
      const float4 q4 = reinterpret_cast<const float4*>(quats)[n];
      q[0] = q4.x; q[1] = q4.y; q[2] = q4.z; q[3] = q4.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) q[i] = quats[4 * n + i];
    }
    load3(scales, aligned16(scales) && coop3, n, row0_3, rows3, lane3, tile3, sc);
    quat_to_rot(q, Rq, qn, inv_norm);
    covar3d(Rq, sc, S);
  }
  const float W = (float)p.width, H = (float)p.height;

  float v_mu[3] = {0.f, 0.f, 0.f};
  float v_S[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // full symmetric gradient matrix, unique entries
  float v_op = 0.f;
  float v_col[(CMODE == GS_COLOR_SH && !SHSPLIT) ? NB * 3 : 4];
#pragma unroll
  for (int k = 0; k < ((CMODE == GS_COLOR_SH && !SHSPLIT) ? NB * 3 : 4); ++k) v_col[k] = 0.f;
  const int n_words = (p.N + 63) >> 6;
  // (the message is laid out for msg_cams >= C cameras: the largest camera count of any rank)
  const MsgLayout ML = msg_layout(SHSPLIT ? msg_cams : p.C, p.N, msg_cap);
  if constexpr (SHSPLIT) {
    if (n_raw == 0) *reinterpret_cast<int32_t*>(msg) = p.C;
  }

  // the radii of up to RCH cameras are fetched together, ahead of the camera loop: one dependent round trip
  // (radius -> branch -> record loads) fewer per camera
  constexpr int RCH = 8;
  int rad[RCH];
  for (int c = 0; c < p.C; ++c) {
    const int64_t idx = (int64_t)c * p.N + n;
    if ((c % RCH) == 0) {
#pragma unroll
      for (int k = 0; k < RCH; ++k) rad[k] = (c + k < p.C) ? radii[(int64_t)(c + k) * p.N + n] : 0;
    }
    int rc = rad[0];
#pragma unroll
    for (int k = 1; k < RCH; ++k) rc = ((c % RCH) == k) ? rad[k] : rc;
    const bool vis = live && (rc > 0);
    int64_t pack_row = 0;  // SHSPLIT: this pair's row of the packed values
    if constexpr (SHSPLIT) {
      const unsigned long long vb = __ballot(vis);
      const int lane = threadIdx.x & 63;
      // (n_raw: the wave's own word even where its tail lanes are clamped to N - 1; a wave wholly past N has none)
      const bool has_word = (n_raw >> 6) < n_words;
      const int64_t wi = (int64_t)c * n_words + (has_word ? (n_raw >> 6) : 0);
      const int32_t base = vis_prefix[wi];
      pack_row = (int64_t)base + __popcll(vb & ((1ull << lane) - 1ull));
      if (lane == 0 && has_word) {
        reinterpret_cast<uint64_t*>(msg + ML.words)[wi] = vb;
        reinterpret_cast<int32_t*>(msg + ML.prefix)[wi] = base;
      }
      if (n_raw == 0) {  // camera centres, once
        const Cam cam0 = load_cam(viewmats, Ks, c);
        float cp0[3];
        cam_center(campos, cam0, c, cp0);
        float* dst = reinterpret_cast<float*>(msg + ML.campos) + 3 * c;
        dst[0] = cp0[0]; dst[1] = cp0[1]; dst[2] = cp0[2];
      }
    }
    float g_R[9], g_t[3], g_cp[3];
    if constexpr (VIEWGRAD) {
#pragma unroll
      for (int i = 0; i < 9; ++i) g_R[i] = 0.f;
#pragma unroll
      for (int i = 0; i < 3; ++i) { g_t[i] = 0.f; g_cp[i] = 0.f; }
    }
    if constexpr (CMODE == GS_COLOR_CN) {
      if (live && !vis)
        for (int k = 0; k < p.color_dim; ++k) v_colors[idx * p.color_dim + k] = 0.f;
    }
    if (vis) {
      const Cam cam = load_cam(viewmats, Ks, c);
      const float4* rec = reinterpret_cast<const float4*>(splats + idx * REC);
      const float4* vrec = reinterpret_cast<const float4*>(v_splats + idx * REC);
      const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
      const float4 g0 = vrec[0], g1 = vrec[1], g2 = vrec[2];
      const float vmx = v_means2d[idx * ld_vm2], vmy = v_means2d[idx * ld_vm2 + 1];
      const float vc0 = g0.z, vc1 = g0.w, vc2 = g1.x;
      const float v_op_eff = g1.y;
      const float vch[4] = {g1.z, g1.w, g2.x, g2.y};
      const float chv[4] = {r1.z, r1.w, r2.x, r2.y};
      const float comp = r3.x;

      // ---- opacity / compensation
      v_op += v_op_eff * comp;
      const float v_comp = p.calc_compensations ? v_op_eff * opac : 0.f;

      // ---- colours
      float v_depth = 0.f;
      int nc = 0;
      float vdx = 0.f, vdy = 0.f, vdz = 0.f;
      if constexpr (CMODE == GS_COLOR_N) {
        nc = p.color_dim;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nc) v_col[k] += vch[k];
      } else if constexpr (CMODE == GS_COLOR_CN) {
        nc = p.color_dim;
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < nc) v_colors[idx * nc + k] = vch[k];
      } else if constexpr (SHSPLIT) {
        nc = 3;
        if (pack_row < msg_cap) {
          float* dst = reinterpret_cast<float*>(msg + ML.vals) + 3 * pack_row;
#pragma unroll
          for (int k = 0; k < 3; ++k) dst[k] = (chv[k] > 0.f) ? vch[k] : 0.f;  // clamp_min(c+0.5, 0) gate
        }
      } else if constexpr (CMODE == GS_COLOR_SH) {
        nc = 3;
        float vrgb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) vrgb[k] = (chv[k] > 0.f) ? vch[k] : 0.f;  // clamp_min(c+0.5, 0) gate
        float cp[3];
        cam_center(campos, cam, c, cp);
        sh_eval_vjp<DEG, true>(colors + (int64_t)n * p.sh_K * 3, mu[0] - cp[0], mu[1] - cp[1], mu[2] - cp[2], vrgb,
                               v_col, vdx, vdy, vdz);
        v_mu[0] += vdx; v_mu[1] += vdy; v_mu[2] += vdz;
        if constexpr (VIEWGRAD) { g_cp[0] = -vdx; g_cp[1] = -vdy; g_cp[2] = -vdz; }
      }
      if (p.depth_channel) {
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k == nc) v_depth = vch[k];
      }

      // ---- recompute forward geometry
      float mc[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) mc[i] = cam.R[i * 3] * mu[0] + cam.R[i * 3 + 1] * mu[1] + cam.R[i * 3 + 2] * mu[2] + cam.t[i];
      float Cc[6];
      covar_world_to_cam(cam.R, S, Cc);
      const Proj2D pr = persp_proj(cam, mc, Cc, W, H, p.eps2d);

      // ---- conic -> blurred cov2d:  v_cov = -X V X, X = conic (full symmetric matrices)
      const float A = r0.z, B = r0.w, Cq = r1.x;
      const float h = 0.5f * vc1;
      const float xv00 = A * vc0 + B * h, xv01 = A * h + B * vc2;
      const float xv10 = B * vc0 + Cq * h, xv11 = B * h + Cq * vc2;
      float s00 = -(xv00 * A + xv01 * B);
      float s01 = -(xv00 * B + xv01 * Cq);
      float s11 = -(xv10 * B + xv11 * Cq);
      if (p.calc_compensations) {
        // add_blur_vjp (Appendix A.1)
        const float det_conic = A * Cq - B * B;
        const float v_sqr = v_comp * 0.5f / (comp + 1e-6f);
        const float om = 1.f - comp * comp;
        s00 += v_sqr * (om * A - p.eps2d * det_conic);
        s01 += v_sqr * (om * B);
        s11 += v_sqr * (om * Cq - p.eps2d * det_conic);
      }

      // ---- persp_proj vjp
      const float x = mc[0], y = mc[1];
      const float rz = pr.rz, rz2 = rz * rz, rz3 = rz2 * rz;
      float v_mc[3];
      v_mc[0] = cam.fx * rz * vmx;
      v_mc[1] = cam.fy * rz * vmy;
      v_mc[2] = -(cam.fx * x * vmx + cam.fy * y * vmy) * rz2;
      // SJ = S2 * J  (2x3):  p = row 0, qv = row 1
      const float p0 = s00 * pr.Ja, p1 = s01 * pr.Jc, p2 = s00 * pr.Jb + s01 * pr.Jd;
      const float q0 = s01 * pr.Ja, q1 = s11 * pr.Jc, q2 = s01 * pr.Jb + s11 * pr.Jd;
      (void)q0;
      // v_Cc = J^T (S2 J)  (full symmetric, unique entries {00,01,02,11,12,22})
      float vC[6];
      vC[0] = pr.Ja * p0; vC[1] = pr.Ja * p1; vC[2] = pr.Ja * p2;
      vC[3] = pr.Jc * q1; vC[4] = pr.Jc * q2; vC[5] = pr.Jb * p2 + pr.Jd * q2;
      // v_J = 2 * (S2 J) Cc
      const float vJ00 = 2.f * (p0 * Cc[0] + p1 * Cc[1] + p2 * Cc[2]);
      const float vJ02 = 2.f * (p0 * Cc[2] + p1 * Cc[4] + p2 * Cc[5]);
      const float vJ11 = 2.f * (q0 * Cc[1] + q1 * Cc[3] + q2 * Cc[4]);
      const float vJ12 = 2.f * (q0 * Cc[2] + q1 * Cc[4] + q2 * Cc[5]);
      if (pr.in_x) v_mc[0] += -cam.fx * rz2 * vJ02; else v_mc[2] += -cam.fx * rz3 * vJ02 * pr.tx;
      if (pr.in_y) v_mc[1] += -cam.fy * rz2 * vJ12; else v_mc[2] += -cam.fy * rz3 * vJ12 * pr.ty;
      v_mc[2] += -cam.fx * rz2 * vJ00 - cam.fy * rz2 * vJ11 + 2.f * cam.fx * pr.tx * rz3 * vJ02 +
                 2.f * cam.fy * pr.ty * rz3 * vJ12;
      v_mc[2] += v_depth;

      // ---- world -> camera vjp
#pragma unroll
      for (int j = 0; j < 3; ++j) v_mu[j] += cam.R[0 * 3 + j] * v_mc[0] + cam.R[1 * 3 + j] * v_mc[1] + cam.R[2 * 3 + j] * v_mc[2];
      // Bm = vC * R ; v_S += R^T Bm
      float Bm[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
          Bm[i * 3 + j] = sym(vC, i, 0) * cam.R[0 * 3 + j] + sym(vC, i, 1) * cam.R[1 * 3 + j] + sym(vC, i, 2) * cam.R[2 * 3 + j];
      auto rtb = [&](int i, int j) { return cam.R[0 * 3 + i] * Bm[0 * 3 + j] + cam.R[1 * 3 + i] * Bm[1 * 3 + j] + cam.R[2 * 3 + i] * Bm[2 * 3 + j]; };
      v_S[0] += rtb(0, 0); v_S[1] += rtb(0, 1); v_S[2] += rtb(0, 2);
      v_S[3] += rtb(1, 1); v_S[4] += rtb(1, 2); v_S[5] += rtb(2, 2);
      if constexpr (VIEWGRAD) {
        // v_R = v_mc (x) mu + 2 vC R Sigma = v_mc (x) mu + 2 Bm Sigma ; v_t = v_mc
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j)
            g_R[i * 3 + j] = v_mc[i] * mu[j] +
                             2.f * (Bm[i * 3 + 0] * sym(S, 0, j) + Bm[i * 3 + 1] * sym(S, 1, j) + Bm[i * 3 + 2] * sym(S, 2, j));
          g_t[i] = v_mc[i];
        }
      }
    }
    if constexpr (VIEWGRAD) {
      // reduce over the wave, one atomic per wave per entry (only when camera poses are optimised:
      // CameraOptModule, /root/reference/3D-Reconstruction-main/models/modules.py:266-311)
      const bool any = __ballot(vis) != 0ull;
      if (any) {
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            const float s = wave_sum(g_R[i * 3 + j]);
            if (lane == 0) atomicAdd(&v_viewmats[c * 16 + i * 4 + j], s);
          }
          const float st = wave_sum(g_t[i]);
          if (lane == 0) atomicAdd(&v_viewmats[c * 16 + i * 4 + 3], st);
          if constexpr (CMODE == GS_COLOR_SH && !SHSPLIT) {
            const float sp = wave_sum(g_cp[i]);
            if (lane == 0 && v_campos) atomicAdd(&v_campos[c * 3 + i], sp);
          }
        }
      }
    }
  }
  if constexpr (CMODE == GS_COLOR_SH && !SHSPLIT) {
    // v_sh [N,K,3]: a lane storing its own 12K-byte row makes every store instruction touch 64 different lines
    // (4x the write requests, measured 0.19 -> 0.17 ms at config 2 and 0.30 -> 0.23 ms at 3 M Gaussians without it).
    // Transpose the wave's rows through LDS, 32 rows at a time, and store them as contiguous 256-B wave rows.
    constexpr int NB3 = NB * 3, TS = NB3 + 1;       // odd row stride: conflict-free
    constexpr int ROWS = (NB3 <= 48) ? 64 : 32;      // rows per round (LDS budget: 4 waves x ROWS x TS floats)
    __shared__ float s_t[4][ROWS * TS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* t = s_t[wave];  // private to the wave
    const int K3 = p.sh_K * 3;
#pragma unroll 1
    for (int h = 0; h < 64 / ROWS; ++h) {
      if (lane / ROWS == h) {
#pragma unroll
        for (int k = 0; k < NB3; ++k) t[(lane % ROWS) * TS + k] = v_col[k];
      }
      const int row0 = n_raw - lane + ROWS * h;  // first Gaussian of this round
      wave_rows_store<NB3>(t, v_colors + (int64_t)row0 * K3, min(ROWS, p.N - row0), K3, lane);
    }
  }
  if (covars) {
    if (!live) return;
    // gradient w.r.t. the 6 upper-triangle parameters: an off-diagonal one sits at (i,j) and (j,i) of Sigma
    v_covars[6 * n + 0] = v_S[0]; v_covars[6 * n + 1] = 2.f * v_S[1]; v_covars[6 * n + 2] = 2.f * v_S[2];
    v_covars[6 * n + 3] = v_S[3]; v_covars[6 * n + 4] = 2.f * v_S[4]; v_covars[6 * n + 5] = v_S[5];
#pragma unroll
    for (int k = 0; k < 3; ++k) v_means[3 * n + k] = v_mu[k];
    v_opacities[n] = v_op;
    if constexpr (CMODE == GS_COLOR_N) {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (k < p.color_dim) v_colors[(int64_t)n * p.color_dim + k] = v_col[k];
    }
    return;
  }

  // ---- Sigma = M M^T, M = Rq diag(s):  v_M = 2 v_S M
  float M[9], vM[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * sc[j];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      vM[i * 3 + j] = 2.f * (sym(v_S, i, 0) * M[0 * 3 + j] + sym(v_S, i, 1) * M[1 * 3 + j] + sym(v_S, i, 2) * M[2 * 3 + j]);
  float vR[9], v_sc[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    v_sc[j] = Rq[0 * 3 + j] * vM[0 * 3 + j] + Rq[1 * 3 + j] * vM[1 * 3 + j] + Rq[2 * 3 + j] * vM[2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 3; ++i) vR[i * 3 + j] = vM[i * 3 + j] * sc[j];
  }
  const float w = qn[0], x = qn[1], y = qn[2], z = qn[3];
  float vq[4];
  vq[0] = 2.f * (x * (vR[7] - vR[5]) + y * (vR[2] - vR[6]) + z * (vR[3] - vR[1]));
  vq[1] = 2.f * (-2.f * x * (vR[4] + vR[8]) + y * (vR[3] + vR[1]) + z * (vR[6] + vR[2]) + w * (vR[7] - vR[5]));
  vq[2] = 2.f * (x * (vR[3] + vR[1]) - 2.f * y * (vR[0] + vR[8]) + z * (vR[7] + vR[5]) + w * (vR[2] - vR[6]));
  vq[3] = 2.f * (x * (vR[6] + vR[2]) + y * (vR[7] + vR[5]) - 2.f * z * (vR[0] + vR[4]) + w * (vR[3] - vR[1]));
  const float dq = vq[0] * w + vq[1] * x + vq[2] * y + vq[3] * z;
  // [N,3] gradients leave as the wave's 768 contiguous bytes (gs_common.hpp, wave_store3); every lane takes part
  store3(v_means, aligned16(v_means) && coop3, n, live, row0_3, rows3, lane3, tile3, v_mu);
  store3(v_scales, aligned16(v_scales) && coop3, n, live, row0_3, rows3, lane3, tile3, v_sc);
  if (!live) return;
  const float4 vq4 = make_float4((vq[0] - dq * qn[0]) * inv_norm, (vq[1] - dq * qn[1]) * inv_norm,
                                 (vq[2] - dq * qn[2]) * inv_norm, (vq[3] - dq * qn[3]) * inv_norm);
  if (aligned16(v_quats)) {
    reinterpret_cast<float4*>(v_quats)[n] = vq4;
  } else {
    v_quats[4 * n] = vq4.x; v_quats[4 * n + 1] = vq4.y; v_quats[4 * n + 2] = vq4.z; v_quats[4 * n + 3] = vq4.w;
  }
  v_opacities[n] = v_op;
  if constexpr (CMODE == GS_COLOR_N) {
#pragma unroll
    for (int k = 0; k < 4; ++k) if (k < p.color_dim) v_colors[(int64_t)n * p.color_dim + k] = v_col[k];
  } else if constexpr (CMODE == GS_COLOR_SH) {
    // (written by the wave-cooperative transposed store above)
  }
}

static int check_params(const gs_proj_params* p) {
  if (!p) return GS_ERR_INVALID_ARG;
  if (p->N < 0 || p->C < 1 || p->width < 1 || p->height < 1 || p->tile_size < 1) return GS_ERR_INVALID_ARG;
  if (p->tile_w != (p->width + p->tile_size - 1) / p->tile_size) return GS_ERR_INVALID_ARG;
  if (p->tile_h != (p->height + p->tile_size - 1) / p->tile_size) return GS_ERR_INVALID_ARG;
  if (p->tile_w >= 65536 || p->tile_h >= 32768) return GS_ERR_UNSUPPORTED;
  if ((int64_t)p->N * p->C >= (int64_t)1 << 31) return GS_ERR_UNSUPPORTED;
  int nc = 0;
  switch (p->color_mode) {
    case GS_COLOR_NONE: nc = 0; break;
    case GS_COLOR_N:
    case GS_COLOR_CN:
      if (p->color_dim < 1 || p->color_dim > 4) return GS_ERR_INVALID_ARG;
      nc = p->color_dim;
      break;
    case GS_COLOR_SH:
      if (p->sh_degree < 0 || p->sh_degree > 4) return GS_ERR_INVALID_ARG;
      if ((p->sh_degree + 1) * (p->sh_degree + 1) > p->sh_K) return GS_ERR_INVALID_ARG;
      nc = 3;
      break;
    default: return GS_ERR_INVALID_ARG;
  }
  if (nc + (p->depth_channel ? 1 : 0) > 4) return GS_ERR_UNSUPPORTED;
  if (nc + (p->depth_channel ? 1 : 0) < 1) return GS_ERR_INVALID_ARG;
  if (p->sh_rows < 0 || p->sh_rows > 2) return GS_ERR_INVALID_ARG;
  return GS_OK;
}

}  // namespace gs

#define GS_FWD_ARGS *p, means, quats, scales, opacities, viewmats, Ks, campos, colors, radii, means2d, splats, tiles_per_gauss, covars, vis_words, vis_counts
#define GS_FWD_LAUNCH(CM, DG) hipLaunchKernelGGL((gs::projection_fwd_kernel<CM, DG>), grid, block, 0, s, GS_FWD_ARGS)
#define GS_FWD_LAUNCH_SH(DG)                                                                                  \
  do {                                                                                                        \
    if (prefetch_rows)                                                                                        \
      hipLaunchKernelGGL((gs::projection_fwd_kernel<GS_COLOR_SH, DG, true>), grid, block, 0, s, GS_FWD_ARGS); \
    else                                                                                                      \
      hipLaunchKernelGGL((gs::projection_fwd_kernel<GS_COLOR_SH, DG, false>), grid, block, 0, s, GS_FWD_ARGS); \
  } while (0)

extern "C" int gs_projection_fwd(const gs_proj_params* p, const float* means, const float* quats,
                                 const float* scales, const float* opacities, const float* viewmats,
                                 const float* Ks, const float* campos, const float* colors, int32_t* radii,
                                 float* means2d, float* splats, int32_t* tiles_per_gauss, const float* covars,
                                 uint64_t* vis_words, int32_t* vis_counts, void* stream) {
  int st = gs::check_params(p);
  if (st != GS_OK) return st;
  if (p->N == 0) return GS_OK;
  if ((vis_words == nullptr) != (vis_counts == nullptr)) return GS_ERR_INVALID_ARG;
  if (!means || !opacities || !viewmats || !Ks || !radii || !splats || !tiles_per_gauss) return GS_ERR_INVALID_ARG;
  if (!covars && (!quats || !scales)) return GS_ERR_INVALID_ARG;
  if (p->color_mode != GS_COLOR_NONE && !colors) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(p->N, 256)), block(256);
  // coefficient rows up front only when most of them will be used: from four cameras on (measured crossover on the
  // 5-camera rig at 1 M Gaussians: 1 camera 0.086 / 0.048 ms prefetched / on demand, 2: 0.090 / 0.076, 3: 0.095 / 0.101,
  // 4: 0.104 / 0.121, 5: 0.111 / 0.127; three neighbouring trajectory views see almost the same Gaussians: on demand)
  const bool prefetch_rows = p->sh_rows == 0 ? p->C >= 4 : p->sh_rows == 1;
  switch (p->color_mode) {
    case GS_COLOR_NONE: GS_FWD_LAUNCH(GS_COLOR_NONE, 0); break;
    case GS_COLOR_N: GS_FWD_LAUNCH(GS_COLOR_N, 0); break;
    case GS_COLOR_CN: GS_FWD_LAUNCH(GS_COLOR_CN, 0); break;
    default:
      switch (p->sh_degree) {
        case 0: GS_FWD_LAUNCH_SH(0); break;
        case 1: GS_FWD_LAUNCH_SH(1); break;
        case 2: GS_FWD_LAUNCH_SH(2); break;
        case 3: GS_FWD_LAUNCH_SH(3); break;
        default: GS_FWD_LAUNCH_SH(4); break;
      }
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

#define GS_BWD_ARGS *p, means, quats, scales, opacities, viewmats, Ks, campos, colors, radii, splats, v_means2d, ld_v_means2d, v_splats, v_means, v_quats, v_scales, v_opacities, v_colors, v_viewmats, v_campos, covars, v_covars, (const int32_t*)nullptr, (unsigned char*)nullptr, (int64_t)0, 0
#define GS_BWD_LAUNCH(CM, DG)                                                                              \
  do {                                                                                                     \
    if (v_viewmats)                                                                                        \
      hipLaunchKernelGGL((gs::projection_bwd_kernel<CM, DG, true>), grid, block, 0, s, GS_BWD_ARGS);       \
    else                                                                                                   \
      hipLaunchKernelGGL((gs::projection_bwd_kernel<CM, DG, false>), grid, block, 0, s, GS_BWD_ARGS);      \
  } while (0)

extern "C" int gs_projection_bwd(const gs_proj_params* p, const float* means, const float* quats,
                                 const float* scales, const float* opacities, const float* viewmats,
                                 const float* Ks, const float* campos, const float* colors,
                                 const int32_t* radii, const float* splats, const float* v_means2d,
                                 int64_t ld_v_means2d, const float* v_splats, float* v_means, float* v_quats,
                                 float* v_scales, float* v_opacities, float* v_colors, float* v_viewmats,
                                 float* v_campos, const float* covars, float* v_covars, void* stream) {
  int st = gs::check_params(p);
  if (st != GS_OK) return st;
  if (p->N == 0) return GS_OK;
  if (!means || !opacities || !viewmats || !Ks || !radii || !splats || !v_means2d || !v_splats || !v_means ||
      !v_opacities || ld_v_means2d < 2)
    return GS_ERR_INVALID_ARG;
  if (covars ? !v_covars : (!quats || !scales || !v_quats || !v_scales)) return GS_ERR_INVALID_ARG;
  if (covars && p->color_mode == GS_COLOR_SH) return GS_ERR_UNSUPPORTED;  // SH colours go through gs_sh_fwd/bwd then
  if (p->color_mode != GS_COLOR_NONE && (!colors || !v_colors)) return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(p->N, 256)), block(256);
  switch (p->color_mode) {
    case GS_COLOR_NONE: GS_BWD_LAUNCH(GS_COLOR_NONE, 0); break;
    case GS_COLOR_N: GS_BWD_LAUNCH(GS_COLOR_N, 0); break;
    case GS_COLOR_CN: GS_BWD_LAUNCH(GS_COLOR_CN, 0); break;
    default:
      switch (p->sh_degree) {
        case 0: GS_BWD_LAUNCH(GS_COLOR_SH, 0); break;
        case 1: GS_BWD_LAUNCH(GS_COLOR_SH, 1); break;
        case 2: GS_BWD_LAUNCH(GS_COLOR_SH, 2); break;
        case 3: GS_BWD_LAUNCH(GS_COLOR_SH, 3); break;
        default: GS_BWD_LAUNCH(GS_COLOR_SH, 4); break;
      }
  }
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// ---- split backward: geometry here, SH colours through the packed colour-gradient message ------------------------
extern "C" int gs_colour_message_layout(int32_t C, int32_t N, int64_t cap, int64_t* out5_host) {
  if (C < 1 || N < 0 || cap < 0 || !out5_host) return GS_ERR_INVALID_ARG;
  const gs::MsgLayout L = gs::msg_layout(C, N, cap);
  out5_host[0] = L.campos; out5_host[1] = L.words; out5_host[2] = L.prefix; out5_host[3] = L.vals; out5_host[4] = L.total;
  return GS_OK;
}

extern "C" int gs_projection_bwd_split(const gs_proj_params* p, const float* means, const float* quats,
                                       const float* scales, const float* opacities, const float* viewmats,
                                       const float* Ks, const float* campos, const int32_t* radii, const float* splats,
                                       const float* v_means2d, int64_t ld_v_means2d, const float* v_splats,
                                       const int32_t* vis_prefix, float* v_means, float* v_quats, float* v_scales,
                                       float* v_opacities, float* v_viewmats, void* message, int64_t cap,
                                       int32_t layout_cams, void* stream) {
  int st = gs::check_params(p);
  if (st != GS_OK) return st;
  if (p->color_mode != GS_COLOR_SH || layout_cams < p->C) return GS_ERR_INVALID_ARG;
  if (p->N == 0) return GS_OK;
  if (!means || !quats || !scales || !opacities || !viewmats || !Ks || !radii || !splats || !v_means2d || !v_splats ||
      !vis_prefix || !v_means || !v_quats || !v_scales || !v_opacities || !message || ld_v_means2d < 2 || cap < 0)
    return GS_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)gs::ceil_div(p->N, 256)), block(256);
  const float* no_f = nullptr;
  float* no_o = nullptr;
#define GS_BWDS_ARGS *p, means, quats, scales, opacities, viewmats, Ks, campos, no_f, radii, splats, v_means2d, ld_v_means2d, v_splats, v_means, v_quats, v_scales, v_opacities, no_o, v_viewmats, no_o, no_f, no_o, vis_prefix, (unsigned char*)message, cap, layout_cams
  if (v_viewmats)
    hipLaunchKernelGGL((gs::projection_bwd_kernel<GS_COLOR_SH, 0, true, true>), grid, block, 0, s, GS_BWDS_ARGS);
  else
    hipLaunchKernelGGL((gs::projection_bwd_kernel<GS_COLOR_SH, 0, false, true>), grid, block, 0, s, GS_BWDS_ARGS);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
