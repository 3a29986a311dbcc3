// Instance-pose transform of the rigid nodes' Gaussians, fused (the part of the
// callers' pre-pass that runs before the activations for RigidNodes).
// The reference (/root/reference/3D-Reconstruction-main/models/nodes/rigid.py:403-470) does, per point with
// instance id p = point_ids[i]:
//   world_mean = quat_to_rotmat(quat_act(instances_quats[frame]))[p] @ mean + instances_trans[frame][p]      (:429-454)
//   world_quat = quat_mult(quat_act(instances_quats[frame][p]), quat_act(quat))                               (:465-470)
// with torch gathers of [n,3,3] / [n,4] / [n,3], a bmm, and -- in the backward -- index_add reductions of n rows onto a
// few dozen instances.  Here: one thread per point; the backward reduces the pose gradients inside the wave per
// distinct instance id (ids are grouped, so usually one or two per wave) before 7 atomics per (wave, instance).
#include "gs_common.hpp"

namespace gs {

__device__ __forceinline__ void quat_rot(const float q[4], float R[9]) {  // unit wxyz -> row-major R
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z); R[1] = 2.f * (x * y - w * z); R[2] = 2.f * (x * z + w * y);
  R[3] = 2.f * (x * y + w * z); R[4] = 1.f - 2.f * (x * x + z * z); R[5] = 2.f * (y * z - w * x);
  R[6] = 2.f * (x * z - w * y); R[7] = 2.f * (y * z + w * x); R[8] = 1.f - 2.f * (x * x + y * y);
}

__device__ __forceinline__ float unit4(const float* __restrict__ src, float q[4]) {
  const float4 v = *reinterpret_cast<const float4*>(src);
  const float n = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w);
  q[0] = v.x / n; q[1] = v.y / n; q[2] = v.z / n; q[3] = v.w / n;
  return n;
}

__global__ __launch_bounds__(256) void rigid_transform_fwd_kernel(int64_t n, int32_t n_inst,
                                                                  const float* __restrict__ means,
                                                                  const float* __restrict__ quats,
                                                                  const int64_t* __restrict__ point_ids,
                                                                  const float* __restrict__ inst_quats,
                                                                  const float* __restrict__ inst_trans,
                                                                  float* __restrict__ o_means, float* __restrict__ o_quats) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t p = point_ids[i];
  float a[4], b[4], R[9];
  unit4(inst_quats + 4 * p, a);
  unit4(quats + 4 * i, b);
  quat_rot(a, R);
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  o_means[3 * i] = R[0] * mx + R[1] * my + R[2] * mz + inst_trans[3 * p];
  o_means[3 * i + 1] = R[3] * mx + R[4] * my + R[5] * mz + inst_trans[3 * p + 1];
  o_means[3 * i + 2] = R[6] * mx + R[7] * my + R[8] * mz + inst_trans[3 * p + 2];
  // Hamilton product a (x) b, models/gaussians/basics.py:64-74
  *reinterpret_cast<float4*>(o_quats + 4 * i) =
      make_float4(a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                  a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ __launch_bounds__(256) void rigid_transform_bwd_kernel(int64_t n, int32_t n_inst,
                                                                  const float* __restrict__ means,
                                                                  const float* __restrict__ quats,
                                                                  const int64_t* __restrict__ point_ids,
                                                                  const float* __restrict__ inst_quats,
                                                                  const float* __restrict__ v_wm,
                                                                  const float* __restrict__ v_wq,
                                                                  float* __restrict__ g_means, float* __restrict__ g_quats,
                                                                  float* __restrict__ g_inst_quats,
                                                                  float* __restrict__ g_inst_trans) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n;
  int p = -1;
  float ga[4] = {0.f, 0.f, 0.f, 0.f}, gt[3] = {0.f, 0.f, 0.f};
  if (live) {
    p = (int)point_ids[i];
    float a[4], b[4], R[9];
    const float na = unit4(inst_quats + 4 * (int64_t)p, a);
    const float nb = unit4(quats + 4 * i, b);
    quat_rot(a, R);
    const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
    const float vx = v_wm ? v_wm[3 * i] : 0.f, vy = v_wm ? v_wm[3 * i + 1] : 0.f, vz = v_wm ? v_wm[3 * i + 2] : 0.f;
    float4 vq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v_wq) vq = *reinterpret_cast<const float4*>(v_wq + 4 * i);
    if (g_means) {  // R^T v
      g_means[3 * i] = R[0] * vx + R[3] * vy + R[6] * vz;
      g_means[3 * i + 1] = R[1] * vx + R[4] * vy + R[7] * vz;
      g_means[3 * i + 2] = R[2] * vx + R[5] * vy + R[8] * vz;
    }
    gt[0] = vx; gt[1] = vy; gt[2] = vz;
    // G = v m^T; vjp of quat_rot at the unit quaternion a
    const float G00 = vx * mx, G01 = vx * my, G02 = vx * mz, G10 = vy * mx, G11 = vy * my, G12 = vy * mz,
                G20 = vz * mx, G21 = vz * my, G22 = vz * mz;
    const float w = a[0], x = a[1], y = a[2], z = a[3];
    float ha[4];
    ha[0] = 2.f * (-z * G01 + y * G02 + z * G10 - x * G12 - y * G20 + x * G21);
    ha[1] = 2.f * (y * G01 + z * G02 + y * G10 - 2.f * x * G11 - w * G12 + z * G20 + w * G21 - 2.f * x * G22);
    ha[2] = 2.f * (-2.f * y * G00 + x * G01 + w * G02 + x * G10 + z * G12 - w * G20 + z * G21 - 2.f * y * G22);
    ha[3] = 2.f * (-2.f * z * G00 - w * G01 + x * G02 + w * G10 - 2.f * z * G11 + y * G12 + x * G20 + y * G21);
    // vjp of the Hamilton product
    ha[0] += vq.x * b[0] + vq.y * b[1] + vq.z * b[2] + vq.w * b[3];
    ha[1] += -vq.x * b[1] + vq.y * b[0] - vq.z * b[3] + vq.w * b[2];
    ha[2] += -vq.x * b[2] + vq.y * b[3] + vq.z * b[0] - vq.w * b[1];
    ha[3] += -vq.x * b[3] - vq.y * b[2] + vq.z * b[1] + vq.w * b[0];
    float hb[4];
    hb[0] = vq.x * a[0] + vq.y * a[1] + vq.z * a[2] + vq.w * a[3];
    hb[1] = -vq.x * a[1] + vq.y * a[0] + vq.z * a[3] - vq.w * a[2];
    hb[2] = -vq.x * a[2] - vq.y * a[3] + vq.z * a[0] + vq.w * a[1];
    hb[3] = -vq.x * a[3] + vq.y * a[2] - vq.z * a[1] + vq.w * a[0];
    // through the normalisations: (h - q (q . h)) / |q_raw|
    const float da = a[0] * ha[0] + a[1] * ha[1] + a[2] * ha[2] + a[3] * ha[3];
    const float db = b[0] * hb[0] + b[1] * hb[1] + b[2] * hb[2] + b[3] * hb[3];
#pragma unroll
    for (int k = 0; k < 4; ++k) ga[k] = (ha[k] - a[k] * da) / na;
    if (g_quats)
      *reinterpret_cast<float4*>(g_quats + 4 * i) = make_float4((hb[0] - b[0] * db) / nb, (hb[1] - b[1] * db) / nb,
                                                                (hb[2] - b[2] * db) / nb, (hb[3] - b[3] * db) / nb);
  }
  // pose gradients: reduce inside the wave per distinct instance id, then 7 atomics per (wave, instance)
  if (!g_inst_quats && !g_inst_trans) return;
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(live);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    const int key = __shfl(p, src);
    const bool mine = live && p == key;
    todo &= ~__ballot(mine);
    float s[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = wave_sum(mine ? ga[k] : 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) s[4 + k] = wave_sum(mine ? gt[k] : 0.f);
    if (lane == src) {
      if (g_inst_quats)
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&g_inst_quats[4 * (int64_t)key + k], s[k]);
      if (g_inst_trans)
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&g_inst_trans[3 * (int64_t)key + k], s[4 + k]);
    }
  }
}

}  // namespace gs

// point_ids [n] int64 in [0, n_inst) (validated by the caller); out_quats is NOT re-normalised (rigid.py:495 does that
// in get_gaussians, i.e. gs_activate_fwd here).
extern "C" int gs_rigid_transform_fwd(int64_t n, int32_t n_inst, const float* means, const float* quats,
                                      const int64_t* point_ids, const float* inst_quats, const float* inst_trans,
                                      float* out_means, float* out_quats, void* stream) {
  if (n < 0 || n_inst < 0) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (n_inst < 1 || !means || !quats || !point_ids || !inst_quats || !inst_trans || !out_means || !out_quats)
    return GS_ERR_INVALID_ARG;
  if (((uintptr_t)quats | (uintptr_t)inst_quats | (uintptr_t)out_quats) & 15) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::rigid_transform_fwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, n_inst, means, quats, point_ids, inst_quats, inst_trans, out_means,
                     out_quats);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// g_inst_quats [n_inst,4] / g_inst_trans [n_inst,3] are accumulated into (zeroed by the caller); any g_* may be NULL,
// v_world_means / v_world_quats may be NULL (= zero).
extern "C" int gs_rigid_transform_bwd(int64_t n, int32_t n_inst, const float* means, const float* quats,
                                      const int64_t* point_ids, const float* inst_quats, const float* v_world_means,
                                      const float* v_world_quats, float* g_means, float* g_quats, float* g_inst_quats,
                                      float* g_inst_trans, void* stream) {
  if (n < 0 || n_inst < 0) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (n_inst < 1 || !means || !quats || !point_ids || !inst_quats) return GS_ERR_INVALID_ARG;
  if (((uintptr_t)quats | (uintptr_t)inst_quats | (uintptr_t)v_world_quats | (uintptr_t)g_quats) & 15)
    return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::rigid_transform_bwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, n_inst, means, quats, point_ids, inst_quats, v_world_means, v_world_quats,
                     g_means, g_quats, g_inst_quats, g_inst_trans);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
