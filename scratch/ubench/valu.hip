// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction at 1..8 waves/SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ __launch_bounds__(64) void k(float* out, long long* cyc, int iters) {
  float2v a[8];
  float s[8];
  for (int i = 0; i < 8; ++i) { a[i] = float2v{(float)threadIdx.x * 1e-3f + i, 1.f}; s[i] = threadIdx.x * 1e-3f + i; }
  float2v b = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
  float bs = 1.0001f, cs = 1e-6f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(bs), "v"(cs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(s[i]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 3) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 4) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[i]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 5) {
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[0]), "+v"(s[1]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[2]), "+v"(s[3]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[4]), "+v"(s[5]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[6]), "+v"(s[7]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[0]), "+v"(s[1]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[2]), "+v"(s[3]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[4]), "+v"(s[5]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[6]), "+v"(s[7]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[0]), "+v"(s[1]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[2]), "+v"(s[3]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[4]), "+v"(s[5]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[6]), "+v"(s[7]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[0]), "+v"(s[1]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[2]), "+v"(s[3]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[4]), "+v"(s[5]));
      asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(s[6]), "+v"(s[7]));
    } else if constexpr (MODE == 6) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 7) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 8) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 9) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(s[i]), "v"(bs) : "vcc");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 10) {
#define X(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(s[i]) : "v"(bs) : "s20", "s21");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 11) {
#define X(i) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(bs) : "vcc");
      REP8(X)
#undef X
    } else if constexpr (MODE == 12) {
#define X(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 13) {
#define X(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(s[i]) : "v"(bs), "v"(cs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 14) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 15) {
#define X(i) asm volatile("v_cmp_le_f32 s[20:21], %0, %1" :: "v"(s[i]), "v"(bs) : "s20", "s21");
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 16) {
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 17) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 18) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, |%0|" : "+v"(s[i]) : "v"(bs));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (MODE == 19) {
#define X(i) asm volatile("v_add_f32 %0, s20, %0" : "+v"(s[i]) :: "s20");
      REP8(X) REP8(X)
#undef X
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y + s[i];
  out[blockIdx.x * 64 + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
  const int iters = 2000;
  for (int wps : {1, 2, 4, 8}) {
    int nblk = 256 * 4 * wps;
    float* out; long long* cyc;
    hipMalloc(&out, nblk * 64 * sizeof(float));
    hipMalloc(&cyc, nblk * sizeof(long long));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<nblk, 64>>>(out, cyc, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<nblk, 64>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(nblk);
    hipMemcpy(h.data(), cyc, nblk * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    double ninst = 16.0 * iters;
    // memtime ticks at 100MHz? report both ticks/inst and wall ns per inst per SIMD
    printf("%-22s waves/SIMD=%d  ticks/inst/wave=%.3f  wall=%.3f ms  ns per SIMD-inst=%.3f (=> %.2f cyc @2.4GHz)\n", name, wps,
           avg / ninst, ms, ms * 1e6 / (ninst * wps), ms * 1e6 / (ninst * wps) * 2.4);
    hipFree(out); hipFree(cyc);
  }
}

int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<6>("v_pk_mul_f32");
  run<8>("v_pk_add_f32");
  run<2>("v_exp_f32");
  run<3>("v_rcp_f32");
  run<4>("v_add_f32_dpp");
  run<5>("v_permlane32_swap");
  run<7>("v_cndmask_b32");
  run<9>("v_cmp_le_f32");
  run<10>("v_cndmask sgpr mask");
  run<11>("cmp+cndmask pair(x8)");
  run<12>("v_max_f32");
  run<13>("v_cndmask nondep");
  run<14>("v_mul_f32");
  run<15>("v_cmp -> sgpr");
  run<16>("v_mov_b32");
  run<17>("v_add_f32");
  run<18>("v_fma_f32 abs mod");
  run<19>("v_add_f32 sgpr src");
  return 0;
}
