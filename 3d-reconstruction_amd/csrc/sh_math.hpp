// Real spherical-harmonics basis (Sloan's efficient evaluation, degree <= 4) and its gradient w.r.t.
// the unit direction.  Semantics: gsplat.cuda._wrapper.spherical_harmonics as called by
// /root/reference/3D-Reconstruction-main/models/gaussians/vanilla.py:369.
#pragma once
#include "gs_common.hpp"

namespace gs {

// basis values b[0..(DEG+1)^2) for a UNIT direction (x,y,z).  Fully unrolled: b[] stays in VGPRs.
template <int DEG>
__device__ __forceinline__ void sh_basis(float x, float y, float z, float* __restrict__ b) {
  b[0] = 0.2820947917738781f;
  if constexpr (DEG >= 1) {
    b[1] = -0.48860251190292f * y;
    b[2] = 0.48860251190292f * z;
    b[3] = -0.48860251190292f * x;
  }
  if constexpr (DEG >= 2) {
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.f * x * y;
    b[4] = 0.5462742152960395f * fS1;
    b[5] = fTmp0B * y;
    b[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    b[7] = fTmp0B * x;
    b[8] = 0.5462742152960395f * fC1;
    if constexpr (DEG >= 3) {
      const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
      const float fTmp1B = 1.445305721320277f * z;
      const float fC2 = x * fC1 - y * fS1;
      const float fS2 = x * fS1 + y * fC1;
      b[9] = -0.5900435899266435f * fS2;
      b[10] = fTmp1B * fS1;
      b[11] = fTmp0C * y;
      b[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
      b[13] = fTmp0C * x;
      b[14] = fTmp1B * fC1;
      b[15] = -0.5900435899266435f * fC2;
      if constexpr (DEG >= 4) {
        const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
        const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
        const float fTmp2B = -1.770130769779931f * z;
        const float fC3 = x * fC2 - y * fS2;
        const float fS3 = x * fS2 + y * fC2;
        b[16] = 0.6258357354491763f * fS3;
        b[17] = fTmp2B * fS2;
        b[18] = fTmp1C * fS1;
        b[19] = fTmp0D * y;
        b[20] = 1.984313483298443f * z * b[12] - 1.006230589874905f * b[6];
        b[21] = fTmp0D * x;
        b[22] = fTmp1C * fC1;
        b[23] = fTmp2B * fC2;
        b[24] = 0.6258357354491763f * fC3;
      }
    }
  }
}

// Gradient of sum_k s[k]*b_k(x,y,z) w.r.t. (x,y,z) treated as independent variables, where s[k] is the
// upstream scalar of basis k.  Returns (gx,gy,gz).
template <int DEG>
__device__ __forceinline__ void sh_basis_vjp(float x, float y, float z, const float* __restrict__ s,
                                             float& gx, float& gy, float& gz) {
  gx = gy = gz = 0.f;
  if constexpr (DEG >= 1) {
    gy += -0.48860251190292f * s[1];
    gz += 0.48860251190292f * s[2];
    gx += -0.48860251190292f * s[3];
  }
  if constexpr (DEG >= 2) {
    const float z2 = z * z;
    const float k2 = -1.092548430592079f;
    const float a = 0.5462742152960395f;
    const float fTmp0B = k2 * z;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.f * x * y;
    // b4 = a*fS1
    gx += s[4] * a * 2.f * y;
    gy += s[4] * a * 2.f * x;
    // b5 = k2*z*y
    gy += s[5] * fTmp0B;
    gz += s[5] * k2 * y;
    // b6
    gz += s[6] * 2.f * 0.9461746957575601f * z;
    // b7 = k2*z*x
    gx += s[7] * fTmp0B;
    gz += s[7] * k2 * x;
    // b8 = a*fC1
    gx += s[8] * a * 2.f * x;
    gy += -s[8] * a * 2.f * y;
    if constexpr (DEG >= 3) {
      const float c0 = -2.285228997322329f;
      const float c1 = 1.445305721320277f;
      const float bb = -0.5900435899266435f;
      const float fTmp0C = c0 * z2 + 0.4570457994644658f;
      const float fTmp1B = c1 * z;
      const float fC2 = x * fC1 - y * fS1;
      const float fS2 = x * fS1 + y * fC1;
      // b9 = bb*fS2
      gx += s[9] * bb * 3.f * fS1;
      gy += s[9] * bb * 3.f * fC1;
      // b10 = fTmp1B*fS1
      gx += s[10] * fTmp1B * 2.f * y;
      gy += s[10] * fTmp1B * 2.f * x;
      gz += s[10] * c1 * fS1;
      // b11 = fTmp0C*y
      gy += s[11] * fTmp0C;
      gz += s[11] * 2.f * c0 * z * y;
      // b12
      const float b12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
      const float b12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
      gz += s[12] * b12_z;
      // b13 = fTmp0C*x
      gx += s[13] * fTmp0C;
      gz += s[13] * 2.f * c0 * z * x;
      // b14 = fTmp1B*fC1
      gx += s[14] * fTmp1B * 2.f * x;
      gy += -s[14] * fTmp1B * 2.f * y;
      gz += s[14] * c1 * fC1;
      // b15 = bb*fC2
      gx += s[15] * bb * 3.f * fC1;
      gy += -s[15] * bb * 3.f * fS1;
      if constexpr (DEG >= 4) {
        const float d0 = -4.683325804901025f, d1 = 2.007139630671868f;
        const float e0 = 3.31161143515146f;
        const float f0 = -1.770130769779931f;
        const float ee = 0.6258357354491763f;
        const float fTmp0D = z * (d0 * z2 + d1);
        const float fTmp0D_z = 3.f * d0 * z2 + d1;
        const float fTmp1C = e0 * z2 - 0.47308734787878f;
        const float fTmp2B = f0 * z;
        // b16 = ee*fS3 ; fS3_x = 4 fS2, fS3_y = 4 fC2
        gx += s[16] * ee * 4.f * fS2;
        gy += s[16] * ee * 4.f * fC2;
        // b17 = fTmp2B*fS2
        gx += s[17] * fTmp2B * 3.f * fS1;
        gy += s[17] * fTmp2B * 3.f * fC1;
        gz += s[17] * f0 * fS2;
        // b18 = fTmp1C*fS1
        gx += s[18] * fTmp1C * 2.f * y;
        gy += s[18] * fTmp1C * 2.f * x;
        gz += s[18] * 2.f * e0 * z * fS1;
        // b19 = fTmp0D*y
        gy += s[19] * fTmp0D;
        gz += s[19] * fTmp0D_z * y;
        // b20 = 1.9843 z b12 - 1.00623 b6
        gz += s[20] * (1.984313483298443f * (b12 + z * b12_z) -
                       1.006230589874905f * 2.f * 0.9461746957575601f * z);
        // b21 = fTmp0D*x
        gx += s[21] * fTmp0D;
        gz += s[21] * fTmp0D_z * x;
        // b22 = fTmp1C*fC1
        gx += s[22] * fTmp1C * 2.f * x;
        gy += -s[22] * fTmp1C * 2.f * y;
        gz += s[22] * 2.f * e0 * z * fC1;
        // b23 = fTmp2B*fC2
        gx += s[23] * fTmp2B * 3.f * fC1;
        gy += -s[23] * fTmp2B * 3.f * fS1;
        gz += s[23] * f0 * fC2;
        // b24 = ee*fC3 ; fC3_x = 4 fC2, fC3_y = -4 fS2
        gx += s[24] * ee * 4.f * fC2;
        gy += -s[24] * ee * 4.f * fS2;
      }
    }
  }
}

// One Gaussian's NB3 = 3 (deg + 1)^2 coefficient floats into registers.  With `vec` (row base 16-B aligned: K % 4 == 0 and
// an aligned tensor) as aligned float4 loads, all issued back to back; otherwise dword loads.  Left to itself the compiler
// merges `coeffs[3 k + c]` into dwordx4 / dwordx3 loads at 4-B-aligned offsets, which stream a third slower
// (scratch/ubench/readbw.hip: 4.5 vs 6.6 TB/s on 192-B rows).
template <int NB3>
__device__ __forceinline__ void sh_load_row(const float* __restrict__ row, bool vec, float* __restrict__ c) {
  if (vec) {
    constexpr int N4 = NB3 / 4;
    const float4* r4 = reinterpret_cast<const float4*>(row);
#pragma unroll
    for (int k = 0; k < N4; ++k) {
      const float4 v = r4[k];
      c[4 * k] = v.x; c[4 * k + 1] = v.y; c[4 * k + 2] = v.z; c[4 * k + 3] = v.w;
    }
#pragma unroll
    for (int k = N4 * 4; k < NB3; ++k) c[k] = row[k];
  } else {
#pragma unroll
    for (int k = 0; k < NB3; ++k) c[k] = row[k];
  }
}

// colour = sum_k b_k(dir/|dir|) * coeffs[k][ch]; coeffs points at this Gaussian's [K][3] block.
template <int DEG>
__device__ __forceinline__ void sh_eval(const float* __restrict__ coeffs, float dx, float dy, float dz,
                                        float out[3]) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
  float b[NB];
  sh_basis<DEG>(dx * inorm, dy * inorm, dz * inorm, b);
  float r = 0.f, g = 0.f, bl = 0.f;
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    r = fmaf(b[k], coeffs[3 * k + 0], r);
    g = fmaf(b[k], coeffs[3 * k + 1], g);
    bl = fmaf(b[k], coeffs[3 * k + 2], bl);
  }
  out[0] = r; out[1] = g; out[2] = bl;
}

// Backward of sh_eval.  v_coeffs_acc[NB*3] += b_k * v_rgb[ch]; if DIRGRAD, (vx,vy,vz) = dL/d(dir) (un-normalised dir).
template <int DEG, bool DIRGRAD>
__device__ __forceinline__ void sh_eval_vjp(const float* __restrict__ coeffs, float dx, float dy, float dz,
                                            const float v_rgb[3], float* __restrict__ v_coeffs_acc,
                                            float& vx, float& vy, float& vz) {
  constexpr int NB = (DEG + 1) * (DEG + 1);
  const float inorm = rsqrtf(dx * dx + dy * dy + dz * dz);
  const float x = dx * inorm, y = dy * inorm, z = dz * inorm;
  float b[NB];
  sh_basis<DEG>(x, y, z, b);
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    v_coeffs_acc[3 * k + 0] = fmaf(b[k], v_rgb[0], v_coeffs_acc[3 * k + 0]);
    v_coeffs_acc[3 * k + 1] = fmaf(b[k], v_rgb[1], v_coeffs_acc[3 * k + 1]);
    v_coeffs_acc[3 * k + 2] = fmaf(b[k], v_rgb[2], v_coeffs_acc[3 * k + 2]);
  }
  vx = vy = vz = 0.f;
  if constexpr (DIRGRAD && DEG >= 1) {
    float s[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k)
      s[k] = coeffs[3 * k + 0] * v_rgb[0] + coeffs[3 * k + 1] * v_rgb[1] + coeffs[3 * k + 2] * v_rgb[2];
    float gx, gy, gz;
    sh_basis_vjp<DEG>(x, y, z, s, gx, gy, gz);
    // through the normalisation n = d/|d|: v_d = (g - (g.n) n) / |d|
    const float gn = gx * x + gy * y + gz * z;
    vx = (gx - gn * x) * inorm;
    vy = (gy - gn * y) * inorm;
    vz = (gz - gn * z) * inorm;
  }
}

// ---- wave-cooperative row transfer ---------------------------------------------------------------------------------
// A lane that stores its own row of K3 = 3K floats (192 B at K = 16) at a K3-float stride makes every store
// instruction of the wave touch 64 different lines.  Instead the wave writes `rows` CONSECUTIVE rows as one contiguous
// block of rows * K3 floats, 256 B per instruction, out of a wave-private LDS tile t[rows][NB3 + 1] (odd stride: the
// per-lane row writes that fill it are bank-conflict free).  Only the first NB3 floats of a row live in the tile
// (the bands in use); the remaining K3 - NB3 floats of every row are stored as zeros.
// LDS operations of one wave complete in order, so a wavefront fence is all the synchronisation needed.
template <int NB3>
__device__ __forceinline__ void wave_rows_store(const float* __restrict__ t, float* __restrict__ base, int rows, int K3,
                                                int lane) {
  constexpr int TS = NB3 + 1;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
  const int total = rows * K3;
  if (K3 == NB3) {
    for (int m = lane; m < total; m += 64) base[m] = t[m + m / NB3];
  } else {
    const float inv_k3 = 1.f / (float)K3;
    for (int m = lane; m < total; m += 64) {
      int g = (int)(((float)m + 0.5f) * inv_k3);
      int k = m - g * K3;
      if (k < 0) { --g; k += K3; } else if (k >= K3) { ++g; k -= K3; }
      base[m] = (k < NB3) ? t[g * TS + k] : 0.f;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

}  // namespace gs
