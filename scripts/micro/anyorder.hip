// anyorder.hip - does hipExtAnyOrderLaunch let consecutive launches of ONE stream overlap on gfx950?  (hip_ext.h says the flag is "not
// supported on GFX9xx boards"; measured instead of assumed.)  64 single-workgroup kernels that each spin ~20 us: serialised they take
// 64 x (20 + boundary) us, overlapped ~20 us.  Also: the same 64 kernels on two streams without events (the free-overlap ceiling).
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(unsigned long long ticks, int* sink) {  // 100 MHz realtime counter
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (ticks == 0xffffffffffffull) *sink = 1;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int* sink; CK(hipMalloc(&sink, 4));
  const int N = 64; const unsigned long long ticks = 2000;  // 20 us
  auto run = [&](int mode) -> double {
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; i++) {
      if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, sink);
      else if (mode == 1) hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, sink);
      else hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, (i & 1) ? s2 : s, ticks, sink);
    }
    hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  for (int rep = 0; rep < 3; rep++)
    printf("rep %d: plain %.1f us | hipExtAnyOrderLaunch %.1f us | two streams, no events %.1f us  (64 x 20 us single-workgroup kernels)\n", rep, run(0), run(1), run(2));
  return 0;
}
