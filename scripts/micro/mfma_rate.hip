// Microbenchmark: sustained issue rate of the two fp16 MFMA shapes on gfx950, from registers, at 1/2/4 waves per SIMD,
// with the shader clock during the run (s_memtime against the 100 MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
__device__ unsigned long long g_ts[4];

template <int SHAPE, int NACC, int ZERO>  // SHAPE 0: 16x16x32, 1: 32x32x16, 2: 32x32x8 (legacy, 4 halves per lane), 3: 16x16x16 (legacy) ; ZERO: operands all zero
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed) {
  h8 a[4], b[2];
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; a[i][e] = ZERO ? (_Float16)0.f : (_Float16)(((int)(s >> 20) - 2048) * (1.f / 2048.f)); }
  for (int i = 0; i < 2; i++) for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; b[i][e] = ZERO ? (_Float16)0.f : (_Float16)(((int)(s >> 20) - 2048) * (1.f / 2048.f)); }
  f4 acc4[NACC]; f16v acc16[NACC];
  for (int i = 0; i < NACC; i++) { acc4[i] = f4{0, 0, 0, 0}; for (int r = 0; r < 16; r++) acc16[i][r] = 0.f; }
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_ts[0] = __builtin_amdgcn_s_memtime(); g_ts[2] = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (SHAPE == 0) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[(i >> 2) & 1], acc4[i], 0, 0, 0);
      else if (SHAPE == 2) { const h4 a4 = {a[i & 3][0], a[i & 3][1], a[i & 3][2], a[i & 3][3]}, b4 = {b[(i >> 2) & 1][0], b[(i >> 2) & 1][1], b[(i >> 2) & 1][2], b[(i >> 2) & 1][3]}; acc16[i] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, acc16[i], 0, 0, 0); }
      else if (SHAPE == 3) { const h4 a4 = {a[i & 3][0], a[i & 3][1], a[i & 3][2], a[i & 3][3]}, b4 = {b[(i >> 2) & 1][0], b[(i >> 2) & 1][1], b[(i >> 2) & 1][2], b[(i >> 2) & 1][3]}; acc4[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc4[i], 0, 0, 0); }
      else acc16[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i & 3], b[(i >> 2) & 1], acc16[i], 0, 0, 0);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_ts[1] = __builtin_amdgcn_s_memtime(); g_ts[3] = __builtin_amdgcn_s_memrealtime(); }
  float r = 0;
  for (int i = 0; i < NACC; i++) { r += acc4[i][0] + acc4[i][3]; for (int q = 0; q < 16; q++) r += acc16[i][q]; }
  if (r == 1234.5678f) out[threadIdx.x] = r;
}

template <int SHAPE, int NACC, int ZERO>
void run(float* out, int bpc, const char* name) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = (SHAPE == 0 || SHAPE == 3) ? 400000 / NACC * 8 : 200000 / NACC * 8;
  k<SHAPE, NACC, ZERO><<<256 * bpc, 256>>>(out, iters / 10, 1u);
  (void)hipEventRecord(e0);
  k<SHAPE, NACC, ZERO><<<256 * bpc, 256>>>(out, iters, 2u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long ts[4]; (void)hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof(ts));
  const double ghz = (double)(ts[1] - ts[0]) / ((double)(ts[3] - ts[2]) * 10.0);
  const double n_mfma_per_simd = (double)bpc * iters * NACC;
  const double flop = 256.0 * bpc * 4 * (double)iters * NACC * 32768.0;
  printf("%-34s waves/SIMD=%d  %7.2f ms  %7.1f TFLOP/s  clock %.3f GHz  %.1f cycles per MFMA per SIMD\n", name, bpc, ms, flop / (ms * 1e-3) / 1e12, ghz,
         ms * 1e6 * ghz / n_mfma_per_simd);
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  for (int bpc = 1; bpc <= 4; bpc *= 2) {
    if (bpc == 1) { run<0, 8, 0>(out, 1, "16x16x32 f16, 8 accumulators"); run<1, 8, 0>(out, 1, "32x32x16 f16, 8 accumulators"); }
    if (bpc == 2) { run<0, 8, 0>(out, 2, "16x16x32 f16, 8 accumulators"); run<1, 8, 0>(out, 2, "32x32x16 f16, 8 accumulators"); }
    if (bpc == 4) { run<0, 8, 0>(out, 4, "16x16x32 f16, 8 accumulators"); run<1, 8, 0>(out, 4, "32x32x16 f16, 8 accumulators"); }
  }
  run<2, 8, 0>(out, 4, "32x32x8 f16 (legacy), 8 acc [flops printed as if x16]");
  run<3, 8, 0>(out, 4, "16x16x16 f16 (legacy), 8 acc [flops as if x32]");
  run<0, 8, 1>(out, 4, "16x16x32 f16, zero operands");
  run<1, 8, 1>(out, 4, "32x32x16 f16, zero operands");
  run<0, 2, 0>(out, 4, "16x16x32 f16, 2 accumulators");
  run<1, 2, 0>(out, 4, "32x32x16 f16, 2 accumulators");
  return 0;
}
