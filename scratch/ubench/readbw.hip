// Read-only / write-only / copy streaming ceilings on gfx950 with hand-written kernels (how far is torch's 3.8 TB/s
// "pure read" from what the part gives?).  hipcc -O3 --offload-arch=gfx950 readbw.hip -o readbw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int U>
__global__ __launch_bounds__(256) void rd(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  float4 acc = make_float4(0, 0, 0, 0);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n4; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < n4; i += stride) { float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;  // never true: keeps the loads alive without a write stream
}

// contiguous chunk per block (each block streams its own range) instead of grid-stride interleaving
template <int U>
__global__ __launch_bounds__(256) void rd_chunk(const float4* __restrict__ x, size_t n4, float* __restrict__ out) {
  const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
  const size_t b0 = (size_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
  float4 acc = make_float4(0, 0, 0, 0);
  size_t i = b0 + threadIdx.x;
  for (; i + (U - 1) * 256 < b1; i += U * 256) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
  for (; i < b1; i += 256) { float4 v = x[i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;
}

__global__ __launch_bounds__(256) void wr(float4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) y[i] = make_float4(1, 2, 3, 4);
}

__global__ __launch_bounds__(256) void cp(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) y[i] = x[i];
}

// per-lane rows: lane reads its own 192-B row (12 float4 at a 192-B lane stride), the projection / SH access pattern
__global__ __launch_bounds__(256) void rd_rows192(const float4* __restrict__ x, size_t rows, float* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float4 acc = make_float4(0, 0, 0, 0);
  const float4* p = x + r * 12;
#pragma unroll
  for (int k = 0; k < 12; ++k) { float4 v = p[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  const float s = acc.x + acc.y + acc.z + acc.w;
  if (s == 123.456f) out[0] = s;
}

// the same rows read the way the compiler merges `coeffs[3 * k + c]` (sh.hip, K = 16): dwordx4 / dwordx3 loads at
// offsets that are only 4-B aligned
struct __attribute__((packed, aligned(4))) U4 { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) U3 { float x, y, z; };
__global__ __launch_bounds__(256) void rd_rows192_unaligned(const float* __restrict__ x, size_t rows, float* __restrict__ out) {
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  const char* p = (const char*)(x + r * 48);
  float acc = 0.f;
  const int o4[9] = {0, 16, 32, 60, 76, 92, 120, 136, 152};
  const int o3[4] = {48, 108, 168, 180};
  U4 a[9]; U3 b[4];
#pragma unroll
  for (int k = 0; k < 9; ++k) a[k] = *(const U4*)(p + o4[k]);
#pragma unroll
  for (int k = 0; k < 4; ++k) b[k] = *(const U3*)(p + o3[k]);
#pragma unroll
  for (int k = 0; k < 9; ++k) acc += a[k].x + a[k].y + a[k].z + a[k].w;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc += b[k].x + b[k].y + b[k].z;
  if (acc == 123.456f) out[0] = acc;
}

// rows + a 12-B side read and a 12-B write per row (the standalone SH forward's whole traffic)
__global__ __launch_bounds__(256) void rd_rows192_rw(const float4* __restrict__ x, const float* __restrict__ d,
                                                     float* __restrict__ o, size_t rows) {
  const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float4 acc = make_float4(d[3 * r], d[3 * r + 1], d[3 * r + 2], 0);
  const float4* p = x + r * 12;
#pragma unroll
  for (int k = 0; k < 12; ++k) { float4 v = p[k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  o[3 * r] = acc.x; o[3 * r + 1] = acc.y; o[3 * r + 2] = acc.z + acc.w;
}

template <typename F>
static double time_ms(F f, int reps = 20) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const size_t bytes = 1152ull << 20;  // 1.2 GB: far beyond the 256 MiB Infinity Cache
  const size_t n4 = bytes / 16;
  float4 *x, *y; float* out;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&out, 256);
  hipMemset(x, 1, bytes); hipMemset(y, 0, bytes);
  for (int blocks : {512, 1024, 2048, 4096, 8192, 16384}) {
    double m4 = time_ms([&] { hipLaunchKernelGGL(rd<4>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
    double m8 = time_ms([&] { hipLaunchKernelGGL(rd<8>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
    double c8 = time_ms([&] { hipLaunchKernelGGL(rd_chunk<8>, dim3(blocks), dim3(256), 0, 0, x, n4, out); });
    double mw = time_ms([&] { hipLaunchKernelGGL(wr, dim3(blocks), dim3(256), 0, 0, y, n4); });
    double mc = time_ms([&] { hipLaunchKernelGGL(cp, dim3(blocks), dim3(256), 0, 0, x, y, n4); });
    printf("blocks %5d: read U4 %.2f TB/s  read U8 %.2f TB/s  read chunk U8 %.2f TB/s  write %.2f TB/s  copy (r+w) %.2f TB/s\n",
           blocks, bytes / m4 / 1e9, bytes / m8 / 1e9, bytes / c8 / 1e9, bytes / mw / 1e9, 2.0 * bytes / mc / 1e9);
  }
  const size_t rows = bytes / 192;
  double mr = time_ms([&] { hipLaunchKernelGGL(rd_rows192, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, x, rows, out); });
  printf("per-lane 192-B rows: %.2f TB/s\n", bytes / mr / 1e9);
  double mu = time_ms([&] { hipLaunchKernelGGL(rd_rows192_unaligned, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, (const float*)x, rows, out); });
  printf("per-lane 192-B rows, 4-B-aligned dwordx4/x3 mix: %.2f TB/s\n", bytes / mu / 1e9);
  double mrw = time_ms([&] { hipLaunchKernelGGL(rd_rows192_rw, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, 0, x, (const float*)y, (float*)y + rows * 3, rows); });
  printf("per-lane 192-B rows + 12 B read + 12 B write per row: %.2f TB/s of the rows' bytes\n", bytes / mrw / 1e9);
  // the same row read at the standalone SH op's size (3 M rows = 576 MB), back to back and with another kernel's
  // 1.2 GB of writes between launches (what the op sees inside a training step)
  {
    const size_t rows3 = 3000000;
    double mb = time_ms([&] { hipLaunchKernelGGL(rd_rows192, dim3((unsigned)((rows3 + 255) / 256)), dim3(256), 0, 0, x, rows3, out); });
    printf("per-lane 192-B rows, 3 M rows back to back: %.2f TB/s\n", rows3 * 192.0 / mb / 1e9);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    double tot = 0;
    for (int i = 0; i < 10; ++i) {
      hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, y, n4);
      hipEventRecord(a);
      hipLaunchKernelGGL(rd_rows192, dim3((unsigned)((rows3 + 255) / 256)), dim3(256), 0, 0, x, rows3, out);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); tot += ms;
    }
    printf("per-lane 192-B rows, 3 M rows, a 1.2 GB write kernel between launches: %.2f TB/s\n", rows3 * 192.0 / (tot / 10) / 1e9);
    // cold-context ceilings: one launch at a time, another kernel over a DIFFERENT 1.2 GB buffer in between
    auto cold = [&](const char* what, auto flush, auto run, double bytes_moved) {
      double t = 0;
      for (int i = 0; i < 10; ++i) {
        flush();
        hipEventRecord(a);
        run();
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); t += ms;
      }
      printf("%s: %.2f TB/s\n", what, bytes_moved / (t / 10) / 1e9);
    };
    auto flush_w = [&] { hipLaunchKernelGGL(wr, dim3(4096), dim3(256), 0, 0, y, n4); };
    auto flush_r = [&] { hipLaunchKernelGGL(rd<8>, dim3(4096), dim3(256), 0, 0, (const float4*)y, n4, out); };
    const size_t n4s = rows3 * 12;
    cold("coalesced read of 576 MB after a 1.2 GB write elsewhere", flush_w,
         [&] { hipLaunchKernelGGL(rd<8>, dim3(8192), dim3(256), 0, 0, x, n4s, out); }, n4s * 16.0);
    cold("coalesced read of 576 MB after a 1.2 GB read elsewhere", flush_r,
         [&] { hipLaunchKernelGGL(rd<8>, dim3(8192), dim3(256), 0, 0, x, n4s, out); }, n4s * 16.0);
    cold("per-lane 192-B rows, 3 M rows, after a 1.2 GB read elsewhere", flush_r,
         [&] { hipLaunchKernelGGL(rd_rows192, dim3((unsigned)((rows3 + 255) / 256)), dim3(256), 0, 0, x, rows3, out); }, rows3 * 192.0);
    cold("coalesced read of 1.2 GB after a 1.2 GB write elsewhere", flush_w,
         [&] { hipLaunchKernelGGL(rd<8>, dim3(8192), dim3(256), 0, 0, x, n4, out); }, n4 * 16.0);
    cold("write of 1.2 GB after a 1.2 GB read elsewhere", [&] { hipLaunchKernelGGL(rd<8>, dim3(4096), dim3(256), 0, 0, x, n4, out); },
         [&] { hipLaunchKernelGGL(wr, dim3(8192), dim3(256), 0, 0, y, n4); }, n4 * 16.0);
    for (size_t rr : {(size_t)1000000, (size_t)2000000, (size_t)4000000, (size_t)6000000}) {
      double m = time_ms([&] { hipLaunchKernelGGL(rd_rows192, dim3((unsigned)((rr + 255) / 256)), dim3(256), 0, 0, x, rr, out); });
      printf("per-lane 192-B rows, %zu rows back to back: %.2f TB/s (%.1f us)\n", rr, rr * 192.0 / m / 1e9, m * 1e3);
    }
  }
  return 0;
}
