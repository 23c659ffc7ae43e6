// Per-CU load rate from an L2/MALL-resident buffer: buffer_load_dwordx4 -> VGPR  vs  buffer_load ... lds (16 B / lane).
// One 256-thread block per CU streams its own 64 KiB slice repeatedly (L2-resident after the first pass).
// build: hipcc --offload-arch=gfx950 -O3 -o cu_load_rate cu_load_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../stable-diffusion.mojo_amd/csrc/lds_dma.h"
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void k(const char* src, int slice_bytes, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const char* base = src + (size_t)blockIdx.x * slice_bytes;
  const rsrc_t r = make_rsrc(base, slice_bytes);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  u4 acc = {0, 0, 0, 0};
  const int chunk = 256 * 16;  // bytes per block-wide load instruction group
  for (int it = 0; it < iters; it++) {
    for (int off = 0; off < slice_bytes; off += chunk * INFLIGHT) {
      if (MODE == 0) {
        u4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, off + j * chunk + tid * 16, 0, 0);
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) acc ^= v[j];
      } else {
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++)
          blds16(r, (unsigned)(off + j * chunk + tid * 16), 0u, smem + (j & 7) * chunk + wave * 1024);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
  if (MODE == 1) acc[0] = *(unsigned*)(smem + tid * 4);
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[tid] = acc[0];
}

template <int MODE, int INFLIGHT>
double run(const char* src, int slice, int blocks, unsigned* sink) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 200;
  k<MODE, INFLIGHT><<<blocks, 256, 8 * 4096>>>(src, slice, 2, sink);
  (void)hipEventRecord(e0);
  k<MODE, INFLIGHT><<<blocks, 256, 8 * 4096>>>(src, slice, iters, sink);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)slice * iters / (ms * 1e-3) / 1e9;  // GB/s per block (= per CU with one block per CU)
}

int main() {
  const int slice = 64 * 1024;
  for (int blocks : {256, 512}) {
    char* src; unsigned* sink;
    (void)hipMalloc(&src, (size_t)slice * blocks); (void)hipMemset(src, 1, (size_t)slice * blocks); (void)hipMalloc(&sink, 4096);
    printf("blocks=%d (x256 threads), 64 KiB slice per block, L2-resident:\n", blocks);
    printf("  buffer_load b128 -> VGPR, 4 in flight: %.1f GB/s per block\n", run<0, 4>(src, slice, blocks, sink));
    printf("  buffer_load b128 -> VGPR, 8 in flight: %.1f GB/s per block\n", run<0, 8>(src, slice, blocks, sink));
    printf("  buffer_load b128 -> LDS,  4 in flight: %.1f GB/s per block\n", run<1, 4>(src, slice, blocks, sink));
    printf("  buffer_load b128 -> LDS,  8 in flight: %.1f GB/s per block\n", run<1, 8>(src, slice, blocks, sink));
    (void)hipFree(src); (void)hipFree(sink);
  }
  return 0;
}
