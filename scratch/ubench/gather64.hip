// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes of the tile pipeline: kernels whose HBM read
// bytes are known by construction, run once each under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum /
// TCC_EA0_RDREQ_32B_sum); profiles/r02_fetch_calibration.txt holds counter x 1024 x 2 (the correction
// profiles/pmc_traffic.py applies) against the true byte count.
//
//   stream        : every lane one float4, fully coalesced                         -> bytes
//   gather64<4>   : every lane one whole 64-B record at a permuted index           -> bytes
//   gather64<3>   : 48 B of each 64-B record (the rasterizer's forward gather)     -> 64-B lines touched = bytes
//   gather64<1>   : 16 B of each 64-B record                                       -> 64-B lines touched = bytes
// Every record is touched exactly once and the 1.2 GB array is 4.5x the Infinity Cache, so "lines touched" is what HBM
// must deliver when a line is the fetch unit; a counter that reads below it counts requested sectors instead.
//   hipcc -O3 --offload-arch=gfx950 scratch/ubench/gather64.hip -o scratch/ubench/gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

__global__ __launch_bounds__(256) void stream(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i < n4; i += (size_t)gridDim.x * 256) { float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) out[0] = acc;
}

// permutation without a table: records visited in the order (i * ODD) mod 2^k, a bijection on [0, 2^k)
template <int Q>
__global__ __launch_bounds__(256) void gather64(const float4* __restrict__ x, unsigned log2n, float* __restrict__ out) {
  const size_t n = (size_t)1 << log2n;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i < n; i += (size_t)gridDim.x * 256) {
    const size_t r = (i * 2654435761ull) & (n - 1);
    const float4* p = x + r * 4;
#pragma unroll
    for (int q = 0; q < Q; ++q) { float4 v = p[q]; acc += v.x + v.y + v.z + v.w; }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
  const unsigned log2n = 24;                 // 16 Mi records of 64 B = 1 GiB
  const size_t bytes = ((size_t)64) << log2n;
  float4* x; float* out;
  if (hipMalloc(&x, bytes) != hipSuccess || hipMalloc(&out, 256) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(x, 1, bytes);
  const char* which = argc > 1 ? argv[1] : "all";
  const int blocks = 8192;
  auto want = [&](const char* n) { return !strcmp(which, "all") || !strcmp(which, n); };
  if (want("stream")) hipLaunchKernelGGL(stream, dim3(blocks), dim3(256), 0, 0, x, bytes / 16, out);
  if (want("g4")) hipLaunchKernelGGL(gather64<4>, dim3(blocks), dim3(256), 0, 0, x, log2n, out);
  if (want("g3")) hipLaunchKernelGGL(gather64<3>, dim3(blocks), dim3(256), 0, 0, x, log2n, out);
  if (want("g1")) hipLaunchKernelGGL(gather64<1>, dim3(blocks), dim3(256), 0, 0, x, log2n, out);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  printf("true bytes: stream %zu, gather64<4> %zu, gather64<3> requested %zu (lines %zu), gather64<1> requested %zu (lines %zu)\n",
         bytes, bytes, bytes / 4 * 3, bytes, bytes / 4, bytes);
  return 0;
}
