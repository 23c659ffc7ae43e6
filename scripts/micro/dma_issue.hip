// dma_issue.hip - what one `buffer_load_dwordx4 ... offen lds` (LDS-DMA, 1 KiB per wave instruction) costs the ISSUING wave, by address
// pattern: (0) lane-linear 1 KiB; (1) 8 rows x 128 B, row pitch 640 B (a conv A tile piece at C = 320); (2) 8 rows x 128 B, pitch 2560 B
// (C = 1280); (3) 16 rows x 64 B, pitch 640 B; (4) 64 rows x 16 B, pitch 640 B.  One 4-wave block per CU (one wave per SIMD), N
// instructions back to back between two s_memtime reads, then one vmcnt(0); the source (8 MiB) is L2-resident after the first pass.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void blds16(rsrc_t r, unsigned voff, unsigned soff, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)nullptr, 0, 0, 0), (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
__global__ __launch_bounds__(256, 1) void k(const char* src, int pattern, int n, unsigned long long* out, int mode, int stream_kb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned voff;
  if (pattern == 0) voff = lane * 16;
  else if (pattern == 1) voff = (lane >> 3) * 640 + (lane & 7) * 16;
  else if (pattern == 2) voff = (lane >> 3) * 2560 + (lane & 7) * 16;
  else if (pattern == 3) voff = (lane >> 2) * 640 + (lane & 3) * 16;
  else voff = lane * 640;
  voff += (blockIdx.x * 4 + wave) * 16384;  // every wave its own 16 KiB window (L2-resident)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7ffffff0, 0x00020000);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; it += 8) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      // stream_kb = 0: the wave re-reads its 16 KiB window (vL1D hits after the first pass); > 0: every instruction reads new lines of a
      // stream_kb-KiB region shared by the workgroups of an XCD-sized group (L2 hits, like weight tiles every CU pulls)
      const unsigned soff = stream_kb ? (unsigned)(((it + u) * 4 + wave) * 1024) % (unsigned)(stream_kb * 1024) : (unsigned)(u * (pattern == 0 ? 1024 : 128));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 16384 + u * 1024), 16, stream_kb ? (unsigned)(lane * 16) : voff, soff, 0, 0);
      if (mode == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // at most 4 in flight: latency-bound cadence
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memtime();
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
int main() {
  const int nb = 256, n = 256;
  char* src; hipMalloc(&src, (size_t)nb * 4 * 16384 + 65536); hipMemset(src, 1, (size_t)nb * 4 * 16384 + 65536);
  unsigned long long* out; hipMalloc(&out, nb * 4 * 2 * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  const char* names[5] = {"lane-linear 1 KiB", "8 rows x 128 B, pitch 640", "8 rows x 128 B, pitch 2560", "16 rows x 64 B, pitch 640", "64 rows x 16 B, pitch 640"};
  for (int mode = 0; mode < 2; mode++)
    for (int p = 0; p < 5; p++) {
      std::vector<unsigned long long> h(nb * 4 * 2);
      for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k, dim3(nb), dim3(256), 65536, 0, src, p, n, out, mode, 0); hipDeviceSynchronize(); }
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      double a = 0, b = 0; for (int i = 0; i < nb * 4; i++) { a += h[2 * i]; b += h[2 * i + 1]; }
      printf("%s | %-28s: issue %.1f ticks per instruction, issue + drain %.1f (256 CUs x 4 waves, %d instructions each)\n", mode ? "<=4 in flight" : "back to back ", names[p], a / (nb * 4) / n, b / (nb * 4) / n, n);
    }
  // the stream every CU pulls through its L2 (lane-linear 1-KiB pieces of a shared region that does not fit the 32 KiB vL1D): what a weight
  // stream can sustain per CU with the whole chip streaming, by instructions allowed in flight per wave
  for (int skb : {2048, 16384})
    for (int mode = 0; mode < 2; mode++) {
      std::vector<unsigned long long> h(nb * 4 * 2);
      const int n2 = 2048;
      for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(k, dim3(nb), dim3(256), 65536, 0, src, 0, n2, out, mode, skb); hipDeviceSynchronize(); }
      hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
      double b = 0; for (int i = 0; i < nb * 4; i++) b += h[2 * i + 1];
      const double ticks = b / (nb * 4) / n2;
      printf("shared %5d-KiB stream, %s: %.1f ticks per 1-KiB instruction per wave = %.1f B/clk per CU (4 waves)\n", skb, mode ? "<= 4 in flight per wave " : "queue-deep (up to 63)   ", ticks, 4096.0 / ticks);
    }
  return 0;
}
