// Sky environment light lookup (second half).
// The reference's EnvLight (/root/reference/3D-Reconstruction-main/models/modules.py:174-203) is one call,
//   light = nvdiffrast.torch.texture(base[None], l, filter_mode='linear', boundary_mode='cube')
// on a learnable cube map base [6,R,R,3] and per-pixel view directions l (rotated by `to_opengl`, modules.py:186,194).
// nvdiffrast is CUDA-only and absent here, so without this op the sky model cannot run on this part at all.
// Semantics restated from nvdiffrast's documentation and the OpenGL cube-map convention it follows (parity unpinned):
//   faces +x,-x,+y,-y,+z,-z; for direction r with major axis ma:  (sc, tc) =
//     +x: (-rz,-ry)  -x: (+rz,-ry)  +y: (+rx,+rz)  -y: (+rx,-rz)  +z: (+rx,-ry)  -z: (-rx,-ry);  s,t = (sc|tc / |ma| + 1) / 2
//   bilinear filter on texel centres; a tap beyond a face edge is folded over the edge onto the adjacent face (seamless
//   edges).  At the eight corners a tap that leaves the face in both directions is clamped in t before folding in s
//   (nvdiffrast drops that tap and renormalises: the two differ only inside the corner texels).
// One thread per direction; the backward scatters w * v_out into the texture gradient with float atomics.
#include "gs_common.hpp"

namespace gs {

// axes of a face: major axis index and sign, and the (axis, sign) pairs that sc and tc run along
struct CubeFace { int ma, ms, sa, ss, ta, ts; };
__device__ __forceinline__ CubeFace cube_face(int f) {
  switch (f) {
    case 0: return {0, +1, 2, -1, 1, -1};
    case 1: return {0, -1, 2, +1, 1, -1};
    case 2: return {1, +1, 0, +1, 2, +1};
    case 3: return {1, -1, 0, +1, 2, -1};
    case 4: return {2, +1, 0, +1, 1, -1};
    default: return {2, -1, 0, -1, 1, -1};
  }
}

__device__ __forceinline__ int face_of(const float p[3]) {
  const float ax = fabsf(p[0]), ay = fabsf(p[1]), az = fabsf(p[2]);
  if (ax >= ay && ax >= az) return p[0] >= 0.f ? 0 : 1;
  if (ay >= az) return p[1] >= 0.f ? 2 : 3;
  return p[2] >= 0.f ? 4 : 5;
}

// texel (ix, iy) of face f, ix / iy possibly one step outside [0, R): linear index into tex [6,R,R] after folding
__device__ __forceinline__ int cube_texel(int f, int ix, int iy, int R) {
  if ((unsigned)ix < (unsigned)R && (unsigned)iy < (unsigned)R) return (f * R + iy) * R + ix;
  if (!((unsigned)ix < (unsigned)R)) iy = min(max(iy, 0), R - 1);  // corner: clamp t, fold s
  const CubeFace F = cube_face(f);
  const float inv = 2.f / (float)R;
  float sc = ((float)ix + 0.5f) * inv - 1.f, tc = ((float)iy + 0.5f) * inv - 1.f, ma = 1.f;
  // fold the overflow over the edge: the coordinate that left the face is clamped to the edge, the overflow comes off
  // the major axis (the point moves down the adjacent face by the same distance)
  if (fabsf(sc) > 1.f) { ma -= fabsf(sc) - 1.f; sc = sc > 0.f ? 1.f : -1.f; }
  else { ma -= fabsf(tc) - 1.f; tc = tc > 0.f ? 1.f : -1.f; }
  float p[3];
  p[F.ma] = (float)F.ms * ma;
  p[F.sa] = (float)F.ss * sc;
  p[F.ta] = (float)F.ts * tc;
  // the folded point lies on the neighbour's plane: its clamped component is the only one of magnitude 1
  const int g = face_of(p);
  const CubeFace G = cube_face(g);
  const float s2 = (float)G.ss * p[G.sa], t2 = (float)G.ts * p[G.ta];
  const int jx = min(max((int)floorf((s2 + 1.f) * 0.5f * (float)R), 0), R - 1);
  const int jy = min(max((int)floorf((t2 + 1.f) * 0.5f * (float)R), 0), R - 1);
  return (g * R + jy) * R + jx;
}

struct CubeTaps { int idx[4]; float w[4]; };

__device__ __forceinline__ CubeTaps cube_taps(const float* __restrict__ dirs, const float* __restrict__ rot, int64_t i,
                                              int R) {
  float d[3] = {dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]};
  if (rot) {  // l @ rot^T (modules.py:194)
    const float x = d[0], y = d[1], z = d[2];
    d[0] = rot[0] * x + rot[1] * y + rot[2] * z;
    d[1] = rot[3] * x + rot[4] * y + rot[5] * z;
    d[2] = rot[6] * x + rot[7] * y + rot[8] * z;
  }
  const int f = face_of(d);
  const CubeFace F = cube_face(f);
  const float ima = 1.f / fabsf(d[F.ma]);
  const float s = ((float)F.ss * d[F.sa] * ima + 1.f) * 0.5f, t = ((float)F.ts * d[F.ta] * ima + 1.f) * 0.5f;
  const float u = s * (float)R - 0.5f, v = t * (float)R - 0.5f;
  // (clamped so that a NaN / zero direction yields NaN weights, never an out-of-range texel)
  const float fu = fminf(fmaxf(floorf(u), -1.f), (float)(R - 1)), fv = fminf(fmaxf(floorf(v), -1.f), (float)(R - 1));
  const int ix = (int)fu, iy = (int)fv;
  const float ax = u - fu, ay = v - fv;
  CubeTaps T;
  T.idx[0] = cube_texel(f, ix, iy, R);         T.w[0] = (1.f - ax) * (1.f - ay);
  T.idx[1] = cube_texel(f, ix + 1, iy, R);     T.w[1] = ax * (1.f - ay);
  T.idx[2] = cube_texel(f, ix, iy + 1, R);     T.w[2] = (1.f - ax) * ay;
  T.idx[3] = cube_texel(f, ix + 1, iy + 1, R); T.w[3] = ax * ay;
  return T;
}

__global__ __launch_bounds__(256) void cube_texture_fwd_kernel(int64_t n, int R, int C, const float* __restrict__ tex,
                                                               const float* __restrict__ dirs,
                                                               const float* __restrict__ rot, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CubeTaps T = cube_taps(dirs, rot, i, R);
  for (int c = 0; c < C; ++c) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) a = fmaf(T.w[k], tex[(int64_t)T.idx[k] * C + c], a);
    out[i * C + c] = a;
  }
}

__global__ __launch_bounds__(256) void cube_texture_bwd_kernel(int64_t n, int R, int C, const float* __restrict__ dirs,
                                                               const float* __restrict__ rot,
                                                               const float* __restrict__ v_out,
                                                               float* __restrict__ v_tex) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const CubeTaps T = cube_taps(dirs, rot, i, R);
  for (int c = 0; c < C; ++c) {
    const float v = v_out[i * C + c];
    if (v == 0.f) continue;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (T.w[k] != 0.f) atomicAdd(&v_tex[(int64_t)T.idx[k] * C + c], T.w[k] * v);
  }
}

}  // namespace gs

// out [n,C] = bilinear, seamless cube-map lookup of tex [6,R,R,C] at dirs [n,3] (any length; optionally rotated by the
// row-major 3x3 `rot` on the device: d' = rot d).  Zero directions are the caller's problem, as in nvdiffrast.
extern "C" int gs_cube_texture_fwd(int64_t n, int32_t R, int32_t C, const float* tex, const float* dirs,
                                   const float* rot, float* out, void* stream) {
  if (n < 0 || R < 1 || R > 16384 || C < 1 || C > 64) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!tex || !dirs || !out) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::cube_texture_fwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, R, C, tex, dirs, rot, out);
  GS_CHECK_LAUNCH();
  return GS_OK;
}

// v_tex [6,R,R,C] += transpose of the lookup applied to v_out [n,C]; the caller zeroes v_tex.
extern "C" int gs_cube_texture_bwd(int64_t n, int32_t R, int32_t C, const float* dirs, const float* rot,
                                   const float* v_out, float* v_tex, void* stream) {
  if (n < 0 || R < 1 || R > 16384 || C < 1 || C > 64) return GS_ERR_INVALID_ARG;
  if (n == 0) return GS_OK;
  if (!dirs || !v_out || !v_tex) return GS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(gs::cube_texture_bwd_kernel, dim3((unsigned)gs::ceil_div(n, 256)), dim3(256), 0,
                     (hipStream_t)stream, n, R, C, dirs, rot, v_out, v_tex);
  GS_CHECK_LAUNCH();
  return GS_OK;
}
