// Microbenchmark: where do the workgroups of one launch land?  Prints (XCC, SE, CU) of every blockIdx for a launch of 512
// workgroups of 256 threads with 64 KiB of LDS each (two per CU), to see which blockIdx pairs share a CU.
// build: hipcc --offload-arch=gfx950 -O3 -o wg_placement wg_placement.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ void k(unsigned* out, unsigned long long* t) {
  extern __shared__ char smem[];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc;
    t[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  // stay resident for a while so that the whole grid is co-resident
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) { }
  if (threadIdx.x == 1) smem[blockIdx.x & 1023] = 1;
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 512;
  unsigned* d; unsigned long long* dt;
  (void)hipMalloc(&d, n * 8); (void)hipMalloc(&dt, n * 8);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  k<<<n, 256, 65536>>>(d, dt);
  k<<<n, 256, 65536>>>(d, dt);
  (void)hipDeviceSynchronize();
  std::vector<unsigned> h(n * 2); std::vector<unsigned long long> ht(n);
  (void)hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  (void)hipMemcpy(ht.data(), dt, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;
  for (int b = 0; b < n; b++) {
    const unsigned hw = h[b * 2], xcc = h[b * 2 + 1] & 0xf;
    const unsigned cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cu[(xcc << 12) | (se << 8) | (sh << 4) | cu_id].push_back(b);
    if (b < 40) printf("block %3d: xcc %u se %u sh %u cu %2u  (t=%llu)\n", b, xcc, se, sh, cu_id, ht[b] - ht[0]);
  }
  printf("distinct CUs used: %zu\n", cu.size());
  int shown = 0;
  std::map<int, int> diff;
  for (auto& kv : cu) {
    if (shown++ < 12) { printf("cu %05x:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
    if (kv.second.size() == 2) diff[kv.second[1] - kv.second[0]]++;
  }
  printf("blockIdx distance between the two blocks of a CU (distance: count):");
  for (auto& kv : diff) printf(" %d:%d", kv.first, kv.second);
  printf("\n");
  return 0;
}
