// stream_rate.hip - what an elementwise pass (read 16 B, write 16 B per lane, like k_gn_apply) sustains on this board, by tensor size and
// access style: the ceiling the GroupNorm-apply / LayerNorm passes are priced against.
//   mode 0: plain loads / stores, one 16-B item per thread per step, U items in flight per thread, grid = one pass (no grid-stride)
//   mode 1: non-temporal stores        mode 2: non-temporal loads and stores        mode 3: read only (sum)    mode 4: write only
//   mode 5: mode 0 + the apply's arithmetic (convert, scale/shift, SiLU through exp2 + rcp, convert)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int U, int MODE>
__global__ __launch_bounds__(256) void k(const h8* __restrict__ x, h8* __restrict__ y, long n, float mu, float r) {
  const long base = ((long)blockIdx.x * U) * 256 + threadIdx.x;
  h8 v[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const long i = base + (long)u * 256;
    if (MODE == 4) { v[u] = h8{1, 2, 3, 4, 5, 6, 7, 8}; continue; }
    if (i < n) v[u] = (MODE == 2) ? __builtin_nontemporal_load(x + i) : x[i];
  }
  if (MODE == 3) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < U; u++) s += (float)v[u][0];
    if (s == 12345.678f) y[0] = v[0];
    return;
  }
#pragma unroll
  for (int u = 0; u < U; u++) {
    const long i = base + (long)u * 256;
    h8 o = v[u];
    if (MODE == 5) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float f = ((float)v[u][j] - mu) * r;
        f = f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));
        o[j] = (_Float16)f;
      }
    }
    if (i < n) { if (MODE == 1 || MODE == 2) __builtin_nontemporal_store(o, y + i); else y[i] = o; }
  }
}
template <int U, int MODE>
static float run(const h8* x, h8* y, long n, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = (int)((n + 256L * U - 1) / (256L * U));
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<U, MODE>), dim3(grid), dim3(256), 0, 0, x, y, n, 0.1f, 1.3f);
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; i++) hipLaunchKernelGGL((k<U, MODE>), dim3(grid), dim3(256), 0, 0, x, y, n, 0.1f, 1.3f);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters * 1e3f;
}
int main() {
  const long sizes_mb[] = {10, 21, 42, 134, 537};
  h8 *x, *y;
  hipMalloc(&x, 600L << 20); hipMalloc(&y, 600L << 20);
  hipMemset(x, 0, 600L << 20); hipMemset(y, 0, 600L << 20);
  printf("%8s | %-22s %-22s %-22s %-22s %-22s %-22s %-22s\n", "MB", "plain U4 us / TB/s", "plain U8", "nt store U4", "nt both U4", "read only U4", "write only U4",
         "plain U4 + SiLU math");
  for (long mb : sizes_mb) {
    const long n = (mb << 20) / 16;
    const int it = mb > 100 ? 10 : 40;
    float t[7] = {run<4, 0>(x, y, n, it), run<8, 0>(x, y, n, it), run<4, 1>(x, y, n, it), run<4, 2>(x, y, n, it), run<4, 3>(x, y, n, it), run<4, 4>(x, y, n, it),
                  run<4, 5>(x, y, n, it)};
    printf("%8ld |", mb);
    for (int j = 0; j < 7; j++) {
      const double bytes = (double)(mb << 20) * ((j == 4 || j == 5) ? 1.0 : 2.0);
      printf(" %8.1f us %6.2f TB/s  ", t[j], bytes / (t[j] * 1e-6) / 1e12);
    }
    printf("\n");
  }
  return 0;
}
