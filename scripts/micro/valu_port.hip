// Microbenchmark (round 6): what do the two waves of a SIMD cost each other on gfx950?  One 512-thread workgroup per CU (waves w and w + 4
// share a SIMD); waves 0-3 run role RA, waves 4-7 role RB, `iters` times; each wave clocks itself with s_memtime.  Printed: shader clocks
// per iteration for each group, alone and side by side - the price list the 8-wave flash attention schedule (kernels_attn8.hip) is built on.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_port valu_port.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

enum Role { IDLE = 0, MFMA28 = 1, EXP64 = 2, MFMA28_EXP56 = 3, CVT32 = 4, FMA64 = 5, MFMA28_EXP28_CVT14 = 6, EXP64_FMA64 = 7, MFMA28_EXP28 = 8, EXP32 = 9,
            MFMA28_EXP32_CVT16 = 10, EXP32_CVT16 = 11, DSREAD14 = 12, MFMA28_FMA56 = 13 };

template <int R>
__device__ __forceinline__ void body(f16v (&acc)[4], float (&e)[16], unsigned (&cv)[8], const h8& a, const h8& b, const char* lds) {
  if constexpr (R == MFMA28) {
#pragma unroll
    for (int i = 0; i < 28; i++) { acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0); }
  } else if constexpr (R == EXP64 || R == EXP32) {
#pragma unroll
    for (int i = 0; i < (R == EXP64 ? 64 : 32); i++) e[i & 15] = __builtin_amdgcn_exp2f(e[i & 15]);
  } else if constexpr (R == MFMA28_EXP56 || R == MFMA28_EXP28 || R == MFMA28_EXP28_CVT14 || R == MFMA28_EXP32_CVT16 || R == MFMA28_FMA56) {
#pragma unroll
    for (int i = 0; i < 28; i++) {
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
      if constexpr (R == MFMA28_FMA56) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[(2 * i) & 15]) : "v"(0.999f), "v"(0.001f));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[(2 * i + 1) & 15]) : "v"(0.999f), "v"(0.001f));
      } else if constexpr (R == MFMA28_EXP32_CVT16) {
        if (i < 16) {
          e[(2 * i) & 15] = __builtin_amdgcn_exp2f(e[(2 * i) & 15]);
          e[(2 * i + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * i + 1) & 15]);
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[i & 7]) : "v"(e[(2 * i) & 15]), "v"(e[(2 * i + 1) & 15]));
        }
      } else {
        e[(2 * i) & 15] = __builtin_amdgcn_exp2f(e[(2 * i) & 15]);
        if constexpr (R == MFMA28_EXP56) e[(2 * i + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * i + 1) & 15]);
        if constexpr (R == MFMA28_EXP28_CVT14) { if (i & 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[i & 7]) : "v"(e[(2 * i) & 15]), "v"(e[(2 * i - 2) & 15])); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (R == CVT32) {
#pragma unroll
    for (int i = 0; i < 32; i++) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[i & 7]) : "v"(e[i & 15]), "v"(e[(i + 1) & 15]));
  } else if constexpr (R == EXP32_CVT16) {
#pragma unroll
    for (int i = 0; i < 16; i++) {
      e[(2 * i) & 15] = __builtin_amdgcn_exp2f(e[(2 * i) & 15]);
      e[(2 * i + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * i + 1) & 15]);
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[i & 7]) : "v"(e[(2 * i) & 15]), "v"(e[(2 * i + 1) & 15]));
    }
  } else if constexpr (R == FMA64) {
#pragma unroll
    for (int i = 0; i < 64; i++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[i & 15]) : "v"(0.999f), "v"(0.001f));
  } else if constexpr (R == EXP64_FMA64) {
#pragma unroll
    for (int i = 0; i < 64; i++) {
      e[i & 7] = __builtin_amdgcn_exp2f(e[i & 7]);
      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(e[8 + (i & 7)]) : "v"(0.999f), "v"(0.001f));
    }
  } else if constexpr (R == DSREAD14) {
#pragma unroll
    for (int i = 0; i < 14; i++) {
      h8 v = *(const h8*)(lds + i * 1024);
      asm volatile("" ::"v"(v));
    }
  }
}

// generalized split of one key tile's work between the two waves of a SIMD (kernels_attn8.hip): role 100 + k = 28 MFMAs with k exponentials
// and k / 2 packed converts spread between them (k <= 56); role 200 + n = n exponentials + n / 2 converts back to back; 300 + k: as 100 + k with
// the VALU work in front of every SECOND MFMA only (coarser interleave); 400 + n: n v_exp_f16; 500 + n: n v_cvt_f16_f32; 600 + n: n v_perm_b32; 700 + n: n v_cndmask
template <int R>
__device__ __forceinline__ void body2(f16v (&acc)[4], float (&e)[16], unsigned (&cv)[8], const h8& a, const h8& b) {
  if constexpr (R >= 100 && R < 200) {
    constexpr int K = R - 100;
#pragma unroll
    for (int i = 0; i < 28; i++) {
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
      for (int pr = 0; pr < K / 2; pr++)
        if ((pr * 28) / (K / 2 > 0 ? K / 2 : 1) == i) {
          e[(2 * pr) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr) & 15]);
          e[(2 * pr + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr + 1) & 15]);
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[pr & 7]) : "v"(e[(2 * pr) & 15]), "v"(e[(2 * pr + 1) & 15]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (R >= 300 && R < 400) {
    constexpr int K = R - 300;
#pragma unroll
    for (int i = 0; i < 28; i += 4) {
#pragma unroll
      for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int pr = 0; pr < K / 2; pr++)
        if ((pr * 7) / (K / 2 > 0 ? K / 2 : 1) == i / 4) {
          e[(2 * pr) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr) & 15]);
          e[(2 * pr + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr + 1) & 15]);
          asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[pr & 7]) : "v"(e[(2 * pr) & 15]), "v"(e[(2 * pr + 1) & 15]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (R >= 200 && R < 300) {
    constexpr int N = R - 200;
#pragma unroll
    for (int pr = 0; pr < N / 2; pr++) {
      e[(2 * pr) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr) & 15]);
      e[(2 * pr + 1) & 15] = __builtin_amdgcn_exp2f(e[(2 * pr + 1) & 15]);
      asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(cv[pr & 7]) : "v"(e[(2 * pr) & 15]), "v"(e[(2 * pr + 1) & 15]));
    }
  } else if constexpr (R >= 400 && R < 500) {
#pragma unroll
    for (int i = 0; i < R - 400; i++) asm volatile("v_exp_f16 %0, %0" : "+v"(cv[i & 7]));
  } else if constexpr (R >= 500 && R < 600) {
#pragma unroll
    for (int i = 0; i < R - 500; i++) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(cv[i & 7]) : "v"(e[i & 15]));
  } else if constexpr (R >= 600 && R < 700) {
#pragma unroll
    for (int i = 0; i < R - 600; i++) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(cv[i & 7]) : "v"(e[i & 15]), "v"(e[(i + 1) & 15]), "v"(0x07060302u));
  } else if constexpr (R >= 700 && R < 800) {
#pragma unroll
    for (int i = 0; i < R - 700; i++) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(cv[i & 7]) : "v"(e[i & 15]), "v"(e[(i + 1) & 15]));
  } else if constexpr (R >= 800 && R < 900) {
#pragma unroll
    for (int i = 0; i < R - 800; i++) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(cv[i & 7]) : "v"(e[i & 15]), "v"(e[(i + 1) & 15]));
  } else if constexpr (R >= 1000 && R < 1100) {  // 28 MFMAs with a separator behind each: does something between two MFMAs open the VALU port for the partner?
    constexpr int SEP = R - 1000;
#pragma unroll
    for (int i = 0; i < 28; i++) {
      acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i & 3], 0, 0, 0);
      if constexpr (SEP == 1) asm volatile("s_nop 0");
      if constexpr (SEP == 2) asm volatile("s_nop 7");
      if constexpr (SEP == 3) asm volatile("v_nop");
      if constexpr (SEP == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(cv[0]));
      if constexpr (SEP == 5) asm volatile("v_mov_b32 %0, %0" : "+v"(cv[1]));
      if constexpr (SEP == 6) { asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); }
      if constexpr (SEP == 7) asm volatile("s_nop 15");
      if constexpr (SEP == 8) { asm volatile("s_nop 15"); asm volatile("s_nop 7"); }
      if constexpr (SEP == 9) asm volatile("s_sleep 1");
      if constexpr (SEP == 10) { asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); }
      if constexpr (SEP == 11) { asm volatile("s_setprio 0"); asm volatile("s_nop 0"); asm volatile("s_setprio 1"); }
      if constexpr (SEP == 12) { asm volatile("s_setprio 0"); asm volatile("s_nop 3"); asm volatile("s_setprio 1"); }
      if constexpr (SEP == 13) { if (i & 1) { asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); } }
      if constexpr (SEP == 14) { asm volatile("s_setprio 0"); asm volatile("s_setprio 2"); }
      if constexpr (SEP == 15) { asm volatile("s_setprio 0"); asm volatile("s_setprio 3"); }
      if constexpr (SEP == 16) { asm volatile("s_setprio 1"); asm volatile("s_setprio 0"); }
      if constexpr (SEP == 17) { asm volatile("s_setprio 0"); }
      if constexpr (SEP == 18) { asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); e[i & 15] = __builtin_amdgcn_exp2f(e[i & 15]); }
      if constexpr (SEP == 19) { e[i & 15] = __builtin_amdgcn_exp2f(e[i & 15]); asm volatile("s_setprio 0"); asm volatile("s_setprio 1"); }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (R >= 900 && R < 1000) {
#pragma unroll
    for (int i = 0; i < R - 900; i++) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(cv[i & 7]) : "v"(0x3c003c00u));
  }
}

template <int RA, int RB, int PRIO_B>
__global__ __launch_bounds__(512, 2) void k(unsigned long long* ticks, float* out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) char smem[16384];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int i = threadIdx.x; i < 16384 / 4; i += 512) ((float*)smem)[i] = seed * i;
  __syncthreads();
  h8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (_Float16)(seed * 0.37f + 0.01f * i + (threadIdx.x & 7) * 0.1f); b[i] = (_Float16)(seed * 0.2f - 0.02f * i + (threadIdx.x & 3) * 0.3f); }
  f16v acc[4];
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) acc[j][i] = 0.f;
  float e[16];
  for (int i = 0; i < 16; i++) e[i] = -seed * 0.5f - 0.01f * i - (threadIdx.x & 63) * 1e-3f;
  unsigned cv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const char* lds = smem + (threadIdx.x & 63) * 16;
  if (PRIO_B && wave >= 4) __builtin_amdgcn_s_setprio(1);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {
    if constexpr (RA != IDLE) for (int it = 0; it < iters; it++) { if constexpr (RA >= 100) body2<RA>(acc, e, cv, a, b); else body<RA>(acc, e, cv, a, b, lds); for (int i = 0; i < 16; i++) e[i] = fminf(e[i], -0.25f) - 0.5f; }
  } else {
    if constexpr (RB != IDLE) for (int it = 0; it < iters; it++) { if constexpr (RB >= 100) body2<RB>(acc, e, cv, a, b); else body<RB>(acc, e, cv, a, b, lds); for (int i = 0; i < 16; i++) e[i] = fminf(e[i], -0.25f) - 0.5f; }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) ticks[blockIdx.x * 8 + wave] = t1 - t0;
  float r = 0;
  for (int j = 0; j < 4; j++) for (int i = 0; i < 16; i++) r += acc[j][i];
  for (int i = 0; i < 16; i++) r += e[i];
  for (int i = 0; i < 8; i++) r += (float)cv[i];
  if (r == 12345.678f) out[threadIdx.x] = r;
}

static const char* role_name[] = {"idle", "28 mfma", "64 exp", "28 mfma + 56 exp", "32 cvt_pk", "64 fma", "28 mfma + 28 exp + 14 cvt", "64 exp + 64 fma", "28 mfma + 28 exp",
                                  "32 exp", "28 mfma + 32 exp + 16 cvt", "32 exp + 16 cvt", "14 ds_read_b128", "28 mfma + 56 fma"};

template <int RA, int RB, int PRIO_B = 0>
void run(unsigned long long* dt, float* out) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<RA, RB, PRIO_B><<<256, 512>>>(dt, out, 10, 1.f);
  (void)hipEventRecord(e0);
  k<RA, RB, PRIO_B><<<256, 512>>>(dt, out, iters, 1.f);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(256 * 8);
  (void)hipMemcpy(h.data(), dt, h.size() * 8, hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < 256; i++) for (int w = 0; w < 8; w++) (w < 4 ? a : b) += (double)h[i * 8 + w];
  a /= 256.0 * 4 * iters; b /= 256.0 * 4 * iters;
  char na[40], nb[40];
  auto nm = [](int r, char* o) { if (r < 100) snprintf(o, 40, "%s", role_name[r]); else snprintf(o, 40, "role %d", r); };
  nm(RA, na); nm(RB, nb);
  printf("A: %-28s B: %-28s prioB=%d  ticks/iter A %7.1f  B %7.1f   wall %7.1f ns/iter\n", na, nb, PRIO_B, a, b, ms * 1e6 / iters);
}

int main(int argc, char** argv) {
  unsigned long long* dt; float* out;
  (void)hipMalloc(&dt, 256 * 8 * 8); (void)hipMalloc(&out, 4096);
  if (argc > 1) {  // second table: instruction prices and the k sweep
    if (argv[1][0] == '4') {
      printf("== 28 MFMAs + separator (6 flip 0-1, 10 two flips, 11 flip with s_nop 0 inside, 12 s_nop 3 inside, 13 flip behind every second, 14 flip 0-2, 15 flip 0-3, 16 flip 1-0, 17 setprio 0 only, 18 flip + own exp, 19 own exp + flip) | X wave\n");
      run<1006, IDLE>(dt, out); run<1010, IDLE>(dt, out); run<1011, IDLE>(dt, out); run<1012, IDLE>(dt, out); run<1013, IDLE>(dt, out);
      run<1006, 264>(dt, out); run<264, 1006>(dt, out); run<1010, 264>(dt, out); run<264, 1010>(dt, out); run<1011, 264>(dt, out); run<264, 1011>(dt, out);
      run<1012, 264>(dt, out); run<264, 1012>(dt, out); run<1013, 264>(dt, out); run<264, 1013>(dt, out); run<1014, 264>(dt, out); run<264, 1014>(dt, out);
      run<1015, 264>(dt, out); run<264, 1015>(dt, out); run<1016, 264>(dt, out); run<264, 1016>(dt, out); run<1017, 264>(dt, out); run<264, 1017>(dt, out);
      run<1018, 236>(dt, out); run<236, 1018>(dt, out); run<1019, 236>(dt, out); run<236, 1019>(dt, out);
      run<1006, 264, 1>(dt, out); run<264, 1006, 1>(dt, out);
      printf("== lighter X waves under the flipping M wave\n");
      run<1006, 248>(dt, out); run<1006, 232>(dt, out); run<1006, EXP64>(dt, out); run<1006, CVT32>(dt, out); run<1006, FMA64>(dt, out); run<1006, DSREAD14>(dt, out);
      return 0;
    }
    if (argv[1][0] == '3') {
      printf("== 28 MFMAs with a separator behind each (1 s_nop 0, 2 s_nop 7, 3 v_nop, 4 s_add, 5 v_mov, 6 setprio flip, 7 s_nop 15, 8 s_nop 15+7, 9 s_sleep 1) | 64 exp + 32 cvt\n");
      run<1001, IDLE>(dt, out); run<1002, IDLE>(dt, out); run<1003, IDLE>(dt, out); run<1005, IDLE>(dt, out); run<1007, IDLE>(dt, out); run<1008, IDLE>(dt, out); run<1009, IDLE>(dt, out);
      run<1001, 264>(dt, out); run<264, 1001>(dt, out); run<1002, 264>(dt, out); run<264, 1002>(dt, out); run<1003, 264>(dt, out); run<264, 1003>(dt, out);
      run<1004, 264>(dt, out); run<264, 1004>(dt, out); run<1005, 264>(dt, out); run<264, 1005>(dt, out); run<1006, 264>(dt, out); run<264, 1006>(dt, out);
      run<1007, 264>(dt, out); run<264, 1007>(dt, out); run<1008, 264>(dt, out); run<264, 1008>(dt, out); run<1009, 264>(dt, out); run<264, 1009>(dt, out);
      run<1002, 264, 1>(dt, out); run<264, 1002, 1>(dt, out); run<1005, 264, 1>(dt, out); run<264, 1005, 1>(dt, out);
      return 0;
    }
    printf("== prices (alone)\n");
    run<264, IDLE>(dt, out); run<432, IDLE>(dt, out); run<464, IDLE>(dt, out); run<532, IDLE>(dt, out); run<564, IDLE>(dt, out); run<632, IDLE>(dt, out); run<732, IDLE>(dt, out);
    run<832, IDLE>(dt, out); run<932, IDLE>(dt, out); run<964, IDLE>(dt, out);
    printf("== one key tile's work split between the waves: M wave = 28 MFMA + k exp + k/2 cvt, X wave = the rest (64 - k exp, 32 - k/2 cvt)\n");
    run<100, 264>(dt, out); run<264, 100>(dt, out); run<100, 264, 1>(dt, out); run<264, 100, 1>(dt, out);
    run<108, 256>(dt, out); run<256, 108>(dt, out); run<108, 256, 1>(dt, out); run<256, 108, 1>(dt, out);
    run<116, 248>(dt, out); run<248, 116>(dt, out); run<116, 248, 1>(dt, out); run<248, 116, 1>(dt, out);
    run<124, 240>(dt, out); run<240, 124>(dt, out); run<124, 240, 1>(dt, out); run<240, 124, 1>(dt, out);
    run<132, 232>(dt, out); run<232, 132>(dt, out); run<132, 232, 1>(dt, out); run<232, 132, 1>(dt, out);
    run<140, 224>(dt, out); run<224, 140>(dt, out); run<140, 224, 1>(dt, out); run<224, 140, 1>(dt, out);
    run<148, 216>(dt, out); run<216, 148>(dt, out);
    printf("== coarser interleave (VALU work behind every fourth MFMA)\n");
    run<332, 232>(dt, out); run<232, 332>(dt, out); run<316, 248>(dt, out); run<248, 316>(dt, out);
    printf("== symmetric\n");
    run<132, 132>(dt, out); run<156, 156>(dt, out); run<164 - 8, 164 - 8, 1>(dt, out);
    return 0;
  }
  printf("== alone (partner idle)\n");
  run<MFMA28, IDLE>(dt, out); run<EXP64, IDLE>(dt, out); run<EXP32, IDLE>(dt, out); run<CVT32, IDLE>(dt, out); run<FMA64, IDLE>(dt, out); run<EXP64_FMA64, IDLE>(dt, out);
  run<EXP32_CVT16, IDLE>(dt, out); run<DSREAD14, IDLE>(dt, out);
  run<MFMA28_EXP28, IDLE>(dt, out); run<MFMA28_EXP56, IDLE>(dt, out); run<MFMA28_EXP28_CVT14, IDLE>(dt, out); run<MFMA28_EXP32_CVT16, IDLE>(dt, out); run<MFMA28_FMA56, IDLE>(dt, out);
  run<IDLE, MFMA28>(dt, out); run<IDLE, EXP64>(dt, out);
  printf("== side by side\n");
  run<MFMA28, MFMA28>(dt, out); run<EXP64, EXP64>(dt, out); run<EXP32, EXP32>(dt, out); run<FMA64, FMA64>(dt, out); run<CVT32, CVT32>(dt, out);
  run<MFMA28, EXP64>(dt, out); run<EXP64, MFMA28>(dt, out); run<MFMA28, EXP64, 1>(dt, out); run<EXP64, MFMA28, 1>(dt, out);
  run<MFMA28, EXP32>(dt, out); run<EXP32, MFMA28>(dt, out); run<MFMA28, EXP32_CVT16>(dt, out); run<EXP32_CVT16, MFMA28>(dt, out);
  run<MFMA28, CVT32>(dt, out); run<CVT32, MFMA28>(dt, out); run<MFMA28, FMA64>(dt, out); run<FMA64, MFMA28>(dt, out);
  run<MFMA28, DSREAD14>(dt, out); run<DSREAD14, MFMA28>(dt, out);
  run<MFMA28_EXP28, MFMA28_EXP28>(dt, out); run<MFMA28_EXP56, MFMA28_EXP56>(dt, out); run<MFMA28_EXP28_CVT14, MFMA28_EXP28_CVT14>(dt, out);
  run<MFMA28_EXP32_CVT16, EXP32_CVT16>(dt, out); run<EXP32_CVT16, MFMA28_EXP32_CVT16>(dt, out);
  run<MFMA28_EXP32_CVT16, EXP32_CVT16, 1>(dt, out); run<EXP32_CVT16, MFMA28_EXP32_CVT16, 1>(dt, out);
  run<MFMA28_EXP32_CVT16, MFMA28_EXP32_CVT16>(dt, out); run<MFMA28_FMA56, MFMA28_FMA56>(dt, out);
  return 0;
}
