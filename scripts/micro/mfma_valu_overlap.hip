// Microbenchmark: can the VALU (incl. transcendental exp2) run under a wave's / another wave's MFMAs on gfx950?
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define ASMOPS \
      if (AOP == 1) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[i]) : "v"(0.999f), "v"(0.001f)); } \
      if (AOP == 2) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_max_f32 %0, %0, %1" : "+v"(e[i]) : "v"(0.5f)); } \
      if (AOP == 3) { _Pragma("unroll") for (int r = 0; r < 2; r++) _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pe[i]) : "v"(pk1), "v"(pk2)); } \
      if (AOP == 4) { _Pragma("unroll") for (int i = 0; i < 8; i++) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(cv[i]) : "v"(e[i]), "v"(e[(i + 1) & 7])); }

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// mode 0: MFMA only; 1: exp only; 2: fma only; 3: MFMA + exp interleaved in every wave; 4: MFMA + fma interleaved;
// 5: even waves MFMA, odd waves exp; 6: even waves MFMA, odd waves fma
// 7/8: v_fma_f32 (asm, not packable) alone / with MFMA; 9/10: v_max_f32 ; 11/12: v_pk_fma_f32 ; 13/14: v_cvt_pkrtz_f16_f32
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
  const int wave = threadIdx.x >> 6;
  h8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed + i); b[i] = (_Float16)(seed * 2 + i); }
  f16v acc0 = {0}, acc1 = {0};
  float e[8];
  for (int i = 0; i < 8; i++) e[i] = seed + 0.01f * i + threadIdx.x * 1e-4f;
  const bool do_mfma = MODE == 0 || MODE == 3 || MODE == 4 || MODE == 8 || MODE == 10 || MODE == 12 || MODE == 14 || ((MODE == 5 || MODE == 6) && !(wave & 1));
  constexpr int AOP = MODE >= 7 ? (MODE - 7) / 2 + 1 : 0;  // 1 fma, 2 max, 3 pk_fma, 4 cvt_pk
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v pe[4]; for (int i = 0; i < 4; i++) pe[i] = f2v{e[2*i], e[2*i+1]};
  const f2v pk1 = {0.999f, 0.999f}, pk2 = {0.001f, 0.001f};
  unsigned cv[8] = {0,0,0,0,0,0,0,0};
  const bool do_exp = MODE == 1 || MODE == 3 || (MODE == 5 && (wave & 1));
  const bool do_fma = MODE == 2 || MODE == 4 || (MODE == 6 && (wave & 1));
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (do_mfma) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
      }
      if (do_exp) {
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = __builtin_amdgcn_exp2f(e[i]);
      }
      if (do_fma) {
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = __builtin_fmaf(e[i], 0.999f, 0.001f);
      }
      ASMOPS
      __builtin_amdgcn_sched_barrier(0);
      if (do_mfma) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
      }
      if (do_exp) {
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = __builtin_amdgcn_exp2f(e[i]);
      }
      if (do_fma) {
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = __builtin_fmaf(e[i], 0.999f, 0.001f);
      }
      ASMOPS
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  for (int i = 0; i < 4; i++) { e[2*i] += pe[i][0]; e[2*i+1] += pe[i][1]; }
  for (int i = 0; i < 8; i++) e[i] += (float)cv[i];
  float r = 0;
  for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
  for (int i = 0; i < 8; i++) r += e[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE>
float run(float* out, int blocks_per_cu, int iters) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<MODE><<<256 * blocks_per_cu, 256>>>(out, 10, 1.f);
  (void)hipEventRecord(e0);
  k<MODE><<<256 * blocks_per_cu, 256>>>(out, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  const int iters = 20000;
  const char* names[] = {"mfma only", "exp only", "fma only", "mfma+exp same wave", "mfma+fma same wave", "mfma waves | exp waves", "mfma waves | fma waves",
                         "v_fma_f32 x64", "mfma + v_fma_f32 x64", "v_max_f32 x64", "mfma + v_max_f32 x64", "v_pk_fma_f32 x64", "mfma + v_pk_fma_f32 x64",
                         "v_cvt_pkrtz x64", "mfma + v_cvt_pkrtz x64"};
  for (int bpc = 1; bpc <= 4; bpc *= 4) {
    float ms[15] = {run<0>(out, bpc, iters), run<1>(out, bpc, iters), run<2>(out, bpc, iters), run<3>(out, bpc, iters),
                   run<4>(out, bpc, iters), run<5>(out, bpc, iters), run<6>(out, bpc, iters), run<7>(out, bpc, iters), run<8>(out, bpc, iters),
                   run<9>(out, bpc, iters), run<10>(out, bpc, iters), run<11>(out, bpc, iters), run<12>(out, bpc, iters), run<13>(out, bpc, iters), run<14>(out, bpc, iters)};
    // per wave per iteration: 8 MFMA (32x32x16), 64 exp or 64 fma
    for (int m = 0; m < 15; m++) {
      const double ns_per_iter = ms[m] * 1e6 / iters;
      printf("waves/SIMD=%d  %-26s %8.3f ms   %7.1f ns/iter (8 MFMA + 64 valu per wave)\n", bpc, names[m], ms[m], ns_per_iter);
    }
  }
  return 0;
}
