// Probe: wave-specialised K loop for the 256x160 tile - 8 compute waves (two groups one barrier apart, 2 per SIMD) that only read
// fragments and multiply, plus NLOAD loader waves (one per SIMD) that issue every LDS-DMA instruction of the block.
// Why: gemm8p.hip's ablations show the three resources of a K tile each need <= 36 us on conv-sized problems but the loop takes
// 62 us - in the ping-pong schedule the "load" half of an interval (fragment reads at the LDS rate shared by four waves + the
// 60-100 cycle issue cost of each LDS-DMA instruction) is longer than the 8-12 MFMAs of the other group.  Taking the DMA issue
// off the compute waves and using 2 phases of 20 MFMAs per K tile (4 barriers instead of 8) should make the intervals MFMA-bound.
// C[m][n] = sum_k A[m][k] * W[n][k], fp16 in, fp32 accumulate, fp16 out.
// build: hipcc --offload-arch=gfx950 -O3 -o gemm_ws gemm_ws.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "../../stable-diffusion.mojo_amd/csrc/lds_dma.h"

typedef _Float16 half_t;
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void keep(const h8& v) { asm volatile("" ::"v"(v)); }
template <int ABL>
__device__ __forceinline__ f4 mma(const h8& w, const h8& a, const f4& c) {
  if constexpr (ABL & 2) { keep(w); keep(a); return c; }
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0);
}
template <int ABL>
__device__ __forceinline__ h8 ldsr(const char* p, const h8& stale) {
  if constexpr (ABL & 4) return stale;
  else return *(const h8*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NLOAD: loader waves (0 = the compute waves issue the DMA themselves) ; STAG: stagger the two compute groups ; PRIO: s_setprio
template <int WGM, int WGN, int FM, int FN, int NLOAD, int STAG, int PRIO, int ABL = 0, int PH = 2>
__global__ __launch_bounds__((WGM * WGN + NLOAD) * 64, (WGM * WGN + NLOAD) / 4) void kws(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                                                       half_t* __restrict__ C, int M, int N, int K, int tiles_n) {
  constexpr int NC = WGM * WGN;  // compute waves
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16, BMw = FM * 16, BNw = FN * 16;
  constexpr int A_INSTR = BM / 8, W_INSTR = BN / 8, NPIECE = A_INSTR + W_INSTR;
  constexpr int NI = NLOAD ? NLOAD : NC;           // waves that issue DMA
  constexpr int PW = (NPIECE + NI - 1) / NI;       // pieces per issuing wave and K tile
  constexpr int TILE_BYTES = (BM + BN) * 128;
  constexpr int FMh = FM / 2;
  static_assert(FM % 2 == 0 && NC == 8 && A_INSTR % NI == 0, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = NLOAD && wave >= NC;
  const int iw = NLOAD ? wave - NC : wave;  // index among the issuing waves
  const int wm = (wave % NC) / WGN, wn = wave % WGN;
  const bool grp1 = STAG && !loader && wave >= NC / 2;
  const int nt = gridDim.x, bid = (blockIdx.x & 7) * (nt >> 3) + (blockIdx.x >> 3);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = K >> 6;
  const int lrow = lane >> 3, cch = (lane & 7) ^ lrow;
  unsigned voff[PW];
#pragma unroll
  for (int i = 0; i < PW; i++) {
    const int j = iw + i * NI;
    const int row = j * 8 + lrow;
    if (i < A_INSTR / NI) voff[i] = ((unsigned)(m0 + row) * (unsigned)K + cch * 8) * 2;
    else voff[i] = ((unsigned)(n0 + row - BM) * (unsigned)K + cch * 8) * 2;
  }
  auto piece = [&](int kt, int slot, int i) {
    const int j = iw + i * NI;
    if (NPIECE % NI != 0 && j >= NPIECE) return;
    if ((ABL & 1) && kt >= 2) return;
    const int nrec = kt < nk ? 0x7ffffff0 : 0;
    char* dst = smem + slot * TILE_BYTES + j * 1024;
    if (i < A_INSTR / NI) blds16(__builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(A), 0, nrec, 0x00020000), voff[i], (unsigned)kt * 128u, dst);
    else blds16(__builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W), 0, nrec, 0x00020000), voff[i], (unsigned)kt * 128u, dst);
  };
  const bool pw_hi = (NPIECE % NI == 0) || iw < (NPIECE % NI);
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lgk0 = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  const bool issuer = NLOAD ? loader : true;
  // prologue: tiles 0 and 1 in flight, tile 0 landed
  if (issuer) {
#pragma unroll
    for (int i = 0; i < PW; i++) piece(0, 0, i);
#pragma unroll
    for (int i = 0; i < PW; i++) piece(1, 1, i);
    if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>();
  }
  bar();

  if (loader) {
    // 2 * PH barriers per K tile, aligned with compute group 0; this wave's share of tile t+2 spread over those intervals
    constexpr int NB = 2 * PH;
    if (PRIO) __builtin_amdgcn_s_setprio(PRIO == 2 ? 2 : 0);
    int slot = 0;
    for (int t = 0; t < nk; t++) {
      const int nslot = slot == 0 ? 2 : slot - 1;
#pragma unroll
      for (int q = 0; q < NB; q++) {
#pragma unroll
        for (int i = q * PW / NB; i < (q + 1) * PW / NB; i++) piece(t + 2, nslot, i);
        if (q == NB - 1) { if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>(); }  // tile t+1 has landed; tile t+2 stays in flight
        bar();
      }
      slot = slot == 2 ? 0 : slot + 1;
    }
    if (STAG) bar();
    wait_vmcnt<0>();
    return;
  }

  f4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; a++)
#pragma unroll
    for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int rsel = lane & 15, key = lane & 7, cq = lane >> 4;
  const int a_rd = (wm * BMw + rsel) * 128, w_rd = BM * 128 + (wn * BNw + rsel) * 128;
  if (grp1) bar();  // group 1 runs one barrier behind group 0
  int slot = 0;
  // PH = 2: one 32-deep k-step per phase (9 fragment reads, FM x FN MFMAs: balanced) ; PH = 1: the whole K tile per phase
  h8 wf[2][FN] = {}, af[2][FM] = {};
  for (int t = 0; t < nk; t++) {
    const char* s = smem + slot * TILE_BYTES;
    const int nslot = slot == 0 ? 2 : slot - 1;
#pragma unroll
    for (int ph = 0; ph < PH; ph++) {
      const bool last = ph == PH - 1;
#pragma unroll
      for (int kk = ph * (2 / PH); kk < (ph + 1) * (2 / PH); kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FN; b++) wf[kk][b] = ldsr<ABL>(s + w_rd + b * 2048 + coff, wf[kk][b]);
#pragma unroll
        for (int a = 0; a < FM; a++) af[kk][a] = ldsr<ABL>(s + a_rd + a * 2048 + coff, af[kk][a]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (!NLOAD) {
#pragma unroll
        for (int i = ph * PW / PH; i < (ph + 1) * PW / PH; i++) piece(t + 2, nslot, i);
        if (last) { if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>(); }
      }
      if (last) lgk0();  // the tile's last reads retire BEFORE the barrier: its slot is refilled right after it
      bar();
      if (!last) lgk0();
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = ph * (2 / PH); kk < (ph + 1) * (2 / PH); kk++)
#pragma unroll
        for (int a = 0; a < FM; a++)
#pragma unroll
          for (int b = 0; b < FN; b++) acc[a][b] = mma<ABL>(wf[kk][b], af[kk][a], acc[a][b]);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      bar();
    }
    slot = slot == 2 ? 0 : slot + 1;
  }
  if (STAG && !grp1) bar();
  wait_vmcnt<0>();
#pragma unroll
  for (int a = 0; a < FM; a++) {
    const int m = m0 + wm * BMw + a * 16 + rsel;
#pragma unroll
    for (int b = 0; b < FN; b++) {
      const int n = n0 + wn * BNw + b * 16 + 4 * cq;
      h4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) o[r] = (half_t)acc[a][b][r];
      if (m < M && n < N) *(h4*)(C + (long long)m * N + n) = o;
    }
  }
}

__global__ void k_ref(const half_t* A, const half_t* W, float* C, int M, int N, int K, int mstep) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y * mstep;
  if (n >= N || m >= M) return;
  float s = 0.f;
  for (int k = 0; k < K; k++) s += (float)A[(long long)m * K + k] * (float)W[(long long)n * K + k];
  C[(long long)blockIdx.y * N + n] = s;
}
__global__ void k_fill(half_t* p, long long n, unsigned seed, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned s = seed ^ (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
  s = s * 1664525u + 1013904223u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
  p[i] = (half_t)(((int)(s >> 8 & 0xffff) - 32768) * (scale / 32768.f));
}

template <int WGM, int WGN, int FM, int FN, int NLOAD, int STAG, int PRIO, int ABL = 0, int PH = 2>
void run(const char* name, int M, int N, int K, half_t* A, half_t* W, half_t* C, float* Cref, int mstep) {
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16, LDS = 3 * (BM + BN) * 128, NT = (WGM * WGN + NLOAD) * 64;
  if (M % BM || N % BN || K % 64 || ((M / BM) * (N / BN)) % 8) { printf("%-52s %6dx%5dx%5d  (shape not tileable)\n", name, M, N, K); return; }
  auto fn = kws<WGM, WGN, FM, FN, NLOAD, STAG, PRIO, ABL, PH>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int tiles_n = N / BN, nt = (M / BM) * tiles_n;
  (void)hipMemset(C, 0, (size_t)M * N * 2);
  fn<<<nt, NT, LDS>>>(A, W, C, M, N, K, tiles_n);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("%-52s launch failed: %s\n", name, hipGetErrorString(e)); exit(1); }
  const int nrow = (M + mstep - 1) / mstep;
  std::vector<half_t> hc((size_t)M * N); std::vector<float> hr((size_t)nrow * N);
  (void)hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
  (void)hipMemcpy(hr.data(), Cref, hr.size() * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0;
  for (int r = 0; r < nrow; r++)
    for (int n = 0; n < N; n++) { const double d = (double)(float)hc[(size_t)r * mstep * N + n] - hr[(size_t)r * N + n]; num += d * d; den += (double)hr[(size_t)r * N + n] * hr[(size_t)r * N + n]; }
  const double rel = sqrt(num / (den > 0 ? den : 1));
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f, tot = 0.f;
  const int rounds = 5, iters = 20;
  for (int r = 0; r < rounds; r++) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < iters; i++) fn<<<nt, NT, LDS>>>(A, W, C, M, N, K, tiles_n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms / iters); tot += ms / iters;
  }
  const double fl = 2.0 * M * N * K;
  printf("%-52s %6dx%5dx%5d  %4d blocks  %8.2f us (best %8.2f)  %7.1f TF (best %7.1f)  rel_l2 %.2e %s\n", name, M, N, K, nt, tot / rounds * 1e3, best * 1e3,
         fl / (tot / rounds * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12, rel, ABL ? "(ablation)" : rel < 2e-3 ? "ok" : "WRONG");
}

int main() {
  struct Shape { int M, N, K; };
  const Shape shapes[] = {{32768, 320, 2880}, {32768, 320, 5760}, {8192, 5120, 640}, {2048, 10240, 1280}, {16384, 2560, 2560}, {8192, 8192, 4096}};
  for (const Shape& s : shapes) {
    const int M = s.M, N = s.N, K = s.K, mstep = 97;
    half_t *A, *W, *C; float* Cref;
    (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2); (void)hipMalloc(&C, (size_t)M * N * 2);
    const int nrow = (M + mstep - 1) / mstep;
    (void)hipMalloc(&Cref, (size_t)nrow * N * 4);
    k_fill<<<(unsigned)(((long long)M * K + 255) / 256), 256>>>(A, (long long)M * K, 1u, 1.f);
    k_fill<<<(unsigned)(((long long)N * K + 255) / 256), 256>>>(W, (long long)N * K, 2u, 0.05f);
    k_ref<<<dim3((N + 255) / 256, nrow), 256>>>(A, W, Cref, M, N, K, mstep);
    (void)hipDeviceSynchronize();
    run<4, 2, 4, 5, 0, 0, 0>("256x160 2-phase, compute waves issue DMA", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 0, 1, 1>("256x160 2-phase + stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 4, 0, 0>("256x160 2-phase, 4 loader waves", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 4, 1, 0>("256x160 2-phase, 4 loader waves + stagger", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 4, 1, 1>("256x160 2-phase, 4 loader waves + stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 4, 1, 2>("256x160 ... loaders at priority 2", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 4, 4, 1, 1>("256x128 2-phase, 4 loader waves + stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 4, 1, 0, 0, 1>("256x160 1-phase (whole K tile), 4 loaders + stagger", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 0, 1, 1, 0, 1>("256x160 1-phase, no loaders, stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    if (s.N == 320) {
      run<4, 2, 4, 5, 4, 1, 0, 1>("  loaders+stagger, ablation: no DMA in the loop", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 2>("  ablation: no MFMA", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 4>("  ablation: no fragment reads", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 5>("  ablation: MFMA + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 6>("  ablation: DMA + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 3>("  ablation: fragment reads + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 4, 1, 0, 7>("  ablation: barriers only", M, N, K, A, W, C, Cref, mstep);
    }
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(Cref);
  }
  return 0;
}
