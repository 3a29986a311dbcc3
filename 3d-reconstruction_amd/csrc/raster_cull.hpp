// Exact (conservative) cull of a splat against a rectangle of pixel centres, shared by the tile rasterizer (quadrant and
// whole-tile staging) and by the binned tile intersection (culled tile lists).
#pragma once
#include "gs_common.hpp"

namespace gs {

// conic pre-scaled so that vis = exp2(pw) with
//   pw(dx, dy) = A dx^2 + 2 Bh dx dy + C dy^2,   (A, Bh, C) = -0.5 * log2(e) * (conic a, b, c),  d = mean - pixel
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float CONIC_SCALE = -0.5f * LOG2E;

// Exact (conservative) cull, done by ONE lane per splat at staging time: an upper bound of pw over the pixel-centre
// rectangle [x0, x0+ext] x [y0, y0+ext] (ext = 7: one 8x8 quadrant, forward; ext = 15: the whole tile, backward).
// pw is a negative-definite quadratic with its maximum (0) at the mean, so the constrained maximum lies on a face
// nearest the mean: maximise the 1-D quadratic on the face dx = clamp(0) and on the face dy = clamp(0) and take the
// larger.  A splat whose best possible alpha in the rectangle is below 1/255 is skipped by every pixel of it anyway
// (A.5: `alpha < 1/255 -> continue`), so dropping it here changes no output bit; `slack` covers the fp32 rounding
// of the per-pixel evaluation.
__device__ __forceinline__ bool rect_may_touch(float mx, float my, float A, float Bh, float C, float opac, float x0,
                                               float y0, float ext) {
  const float dxh = mx - x0, dxl = dxh - ext;
  const float dyh = my - y0, dyl = dyh - ext;
  const float cx = fminf(fmaxf(0.f, dxl), dxh);
  const float cy = fminf(fmaxf(0.f, dyl), dyh);
  const float ys = fminf(fmaxf(-Bh * cx * __builtin_amdgcn_rcpf(C), dyl), dyh);
  const float xs = fminf(fmaxf(-Bh * cy * __builtin_amdgcn_rcpf(A), dxl), dxh);
  const float pX = A * cx * cx + 2.f * Bh * cx * ys + C * ys * ys;
  const float pY = A * xs * xs + 2.f * Bh * xs * cy + C * cy * cy;
  const float Dx = fmaxf(fabsf(dxl), fabsf(dxh)), Dy = fmaxf(fabsf(dyl), fabsf(dyh));
  const float slack = 1e-5f * (fabsf(A) * Dx * Dx + fabsf(C) * Dy * Dy) + 1e-2f;
  const float pmax = fminf(0.f, fmaxf(pX, pY) + slack);
  // NaN/inf in a record (never produced by the projection for a visible row) -> keep, the pixel loop decides
  return !(opac * __builtin_amdgcn_exp2f(pmax) < ALPHA_MIN * 0.999f);
}

}  // namespace gs
