// Micro-test: does buffer_load_dwordx4 ... offen lds write ZEROS to LDS for out-of-range lanes, and is soffset
// part of the range check?  (Used to fold conv zero padding into the buffer descriptor's bounds check.)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void k(const _Float16* A, int nbytes, const unsigned* offs, unsigned soff, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 1024 / 4; i += 64) ((float*)smem)[i] = -7.f;  // poison
  __syncthreads();
  auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, nbytes, 0x00020000);
  unsigned voff = offs[threadIdx.x];
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)smem, 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 8; i++) out[threadIdx.x * 8 + i] = (float)((_Float16*)smem)[threadIdx.x * 8 + i];
}
int main() {
  const int n = 1024;  // halves
  std::vector<_Float16> h(n);
  for (int i = 0; i < n; i++) h[i] = (_Float16)(i + 1);
  _Float16* dA; unsigned* dO; float* dout;
  hipMalloc(&dA, 4096); hipMemset(dA, 0x3c, 4096);  // bytes beyond the descriptor hold 0x3c3c (=1.0586)
  hipMemcpy(dA, h.data(), n * 2, hipMemcpyHostToDevice);
  std::vector<unsigned> offs(64);
  for (int l = 0; l < 64; l++) offs[l] = l * 16;
  offs[3] = 0xFFFFFFF0u; offs[4] = 0x80000000u; offs[5] = n * 2 - 16; offs[6] = n * 2 - 8; offs[7] = n * 2; offs[8] = n * 2 - 144;
  hipMalloc(&dO, 256); hipMemcpy(dO, offs.data(), 256, hipMemcpyHostToDevice);
  hipMalloc(&dout, 64 * 8 * 4);
  for (unsigned soff : {0u, 128u}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 1024, 0, dA, n * 2, dO, soff, dout);
    std::vector<float> o(512);
    hipMemcpy(o.data(), dout, 2048, hipMemcpyDeviceToHost);
    printf("soffset=%u\n", soff);
    for (int l : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9}) {
      printf("  lane %d voff=0x%x:", l, offs[l]);
      for (int i = 0; i < 8; i++) printf(" %g", o[l * 8 + i]);
      printf("\n");
    }
  }
  return 0;
}
