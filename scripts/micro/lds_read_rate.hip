// ds_read_b128 bytes per clock per CU at 1, 2 and 4 waves per SIMD (conflict-free, swizzled 128-B rows as in the GEMM
// tiles).  Settles whether the LDS port serves 128 or 256 B/clk/CU for the fragment reads (MI355X_MICROARCH.md LDS table
// says 256; DESIGN.md 7b of round 2 priced it at 128).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_read_rate lds_read_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int NREAD>
__global__ void k(int iters, unsigned* sink, unsigned long long* ticks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 64 * 1024 / 4; i += blockDim.x) ((unsigned*)smem)[i] = i * 2654435761u;
  __syncthreads();
  // fragment-read pattern of kernels_gemm.hip: row = lane & 15, 16-B chunk (lane >> 4) ^ (row & 7), 128-B rows
  const int row = lane & 15, key = lane & 7, cq = lane >> 4;
  const char* base = smem + ((wave & 3) * 16 + row) * 128;
  u4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    u4 v[NREAD];
#pragma unroll
    for (int j = 0; j < NREAD; j++) {
      const int coff = (((j & 1) * 4 + cq) ^ key) << 4;
      v[j] = *(const u4*)(base + (j >> 1) * 8192 + coff);
    }
#pragma unroll
    for (int j = 0; j < NREAD; j++) acc ^= v[j];
    asm volatile("" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[tid] = acc[0];
}

template <int NREAD>
void run(int waves_per_simd, unsigned* sink, unsigned long long* dticks) {
  const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int blocks_per_cu = (256 * waves_per_simd) / threads;
  const int blocks = 256 * blocks_per_cu, iters = 4000;
  const int lds = 64 * 1024;
  (void)hipFuncSetAttribute((const void*)k<NREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<NREAD><<<blocks, threads, lds>>>(10, sink, dticks);
  (void)hipEventRecord(e0);
  k<NREAD><<<blocks, threads, lds>>>(iters, sink, dticks);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; (void)hipMemcpy(h, dticks, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes_cu = (double)waves_per_simd * 4 * 64 * 16 * NREAD * iters;  // per CU
  printf("  %d wave(s)/SIMD, %2d ds_read_b128 per wait: %6.1f B/clk/CU (s_memtime), %6.1f TB/s chip (wall %.3f ms)\n", waves_per_simd, NREAD,
         bytes_cu / (double)h[0], bytes_cu * 256 / (ms * 1e-3) / 1e12, ms);
}

int main() {
  unsigned* sink; unsigned long long* dt;
  (void)hipMalloc(&sink, 4096); (void)hipMalloc(&dt, 8192 * 8);
  printf("ds_read_b128, conflict-free swizzled fragment pattern, every CU busy:\n");
  for (int w : {1, 2, 4}) { run<8>(w, sink, dt); run<16>(w, sink, dt); }
  return 0;
}
