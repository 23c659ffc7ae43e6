// Standalone probe of the "8-phase" K loop (cdna_hip_programming.md, "The 256^2 8-phase template") on THIS repo's shapes:
// 8 waves (2 per SIMD) in two groups that run one barrier apart - while one group multiplies a C quadrant the other reads the
// next quadrant's fragments from LDS and issues LDS-DMA - four quadrant phases per 64-deep K tile, s_setprio around the MFMA
// groups, whole K tiles DMA'd two ahead into a 3-slot ring behind ONE counted s_waitcnt vmcnt per K tile.
// C[m][n] = sum_k A[m][k] * W[n][k], fp16 in, fp32 accumulate, fp16 out (same operand layout as kernels_gemm.hip).
// build: hipcc --offload-arch=gfx950 -O3 -o gemm8p gemm8p.hip
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
template <int MODE>
__device__ __forceinline__ f4 mma(const h8& w, const h8& a, const f4& c) {
  if constexpr (MODE & 16) { keep(w); keep(a); return c; }
  else return __builtin_amdgcn_mfma_f32_16x16x32_f16(w, a, c, 0, 0, 0);
}
template <int MODE>
__device__ __forceinline__ h8 ldsr(const char* p, const h8& stale) {
  if constexpr (MODE & 32) return stale;
  else return *(const h8*)p;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// MODE bit 0: stagger the two wave groups by one barrier (ping-pong) ; bit 1: s_setprio around the MFMA groups ;
// bit 2: lockstep baseline = all fragment reads of the K tile, then all MFMAs, one barrier per tile (cfg 11 of kernels_gemm.hip)
template <int WGM, int WGN, int FM, int FN, int MODE>
__global__ __launch_bounds__(WGM* WGN * 64, (WGM * WGN) / 4) void k8p(const half_t* __restrict__ A, const half_t* __restrict__ W,
                                                                     half_t* __restrict__ C, int M, int N, int K, int tiles_n) {
  constexpr int NW = WGM * WGN, NS = 3;
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16, BMw = FM * 16, BNw = FN * 16;
  constexpr int A_INSTR = BM / 8, W_INSTR = BN / 8, NPIECE = A_INSTR + W_INSTR;
  constexpr int PW = (NPIECE + NW - 1) / NW;  // DMA pieces per wave and K tile (waves below NPIECE % NW issue PW, the rest PW - 1)
  constexpr int TILE_BYTES = (BM + BN) * 128;
  constexpr int FNa = (FN + 1) / 2, FNb = FN - FNa, FMh = FM / 2;
  static_assert(FM % 2 == 0 && NW == 8, "8 waves, even FM");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const bool grp1 = (MODE & 1) && wave >= NW / 2;
  // XCD-aware tile map (block b runs on XCD b % 8): each XCD owns a contiguous range of tiles
  const int nt = gridDim.x, bid = (blockIdx.x & 7) * (nt >> 3) + (blockIdx.x >> 3);
  const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = K >> 6;
  const int lrow = lane >> 3, cch = (lane & 7) ^ lrow;
  // piece j of a tile: rows j*8 .. j*8+7 of [A tile ; W tile]
  unsigned voff[PW];
#pragma unroll
  for (int i = 0; i < PW; i++) {
    const int j = wave + i * NW;
    const int row = j * 8 + lrow;
    if (j < A_INSTR) voff[i] = ((unsigned)(m0 + row) * (unsigned)K + cch * 8) * 2;
    else voff[i] = ((unsigned)(n0 + row - BM) * (unsigned)K + cch * 8) * 2;
  }
  static_assert(A_INSTR % NW == 0, "the A / W split of a wave's pieces must be a compile-time property of the piece index");
  auto piece = [&](int kt, int slot, int i) {  // i: compile-time index of this wave's piece (A pieces first)
    const int j = wave + i * NW;
    if (NPIECE % NW != 0 && j >= NPIECE) return;
    if ((MODE & 8) && kt >= 2) return;  // ablation: no DMA inside the loop
    // descriptors are rebuilt from scalars (a dead tile = zero records: the DMA writes zeros): a run-time SELECT between two
    // descriptors lands in VGPRs and hipcc wraps every DMA in a waterfall loop behind s_waitcnt vmcnt(0)
    const int nrec = kt < nk ? 0x7ffffff0 : 0;
    char* dst = smem + slot * TILE_BYTES + j * 1024;
    if (i < A_INSTR / NW) blds16(__builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(A), 0, nrec, 0x00020000), voff[i], (unsigned)kt * 128u, dst);
    else blds16(__builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(W), 0, nrec, 0x00020000), voff[i], (unsigned)kt * 128u, dst);
  };
  const bool pw_hi = (NPIECE % NW == 0) || wave < (NPIECE % NW);

  f4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; a++)
#pragma unroll
    for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int rsel = lane & 15, key = lane & 7, cq = lane >> 4;
  const int a_rd = (wm * BMw + rsel) * 128, w_rd = BM * 128 + (wn * BNw + rsel) * 128;

  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  auto lgk0 = [&]() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: tiles 0 and 1 in flight, tile 0 landed
#pragma unroll
  for (int i = 0; i < PW; i++) piece(0, 0, i);
#pragma unroll
  for (int i = 0; i < PW; i++) piece(1, 1, i);
  if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>();
  bar();
  if (MODE & 8) { wait_vmcnt<0>(); bar(); }

  if constexpr (MODE & 4) {
    // lockstep baseline: one barrier per K tile, all fragments then all MFMAs, next-next tile's DMA in the MFMA shadow
    int slot = 0;
    for (int t = 0; t < nk; t++) {
      const char* s = smem + slot * TILE_BYTES;
      h8 af[2][FM], wf[2][FN];
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FN; b++) wf[kk][b] = *(const h8*)(s + w_rd + b * 2048 + coff);
#pragma unroll
        for (int a = 0; a < FM; a++) af[kk][a] = *(const h8*)(s + a_rd + a * 2048 + coff);
      }
      __builtin_amdgcn_sched_barrier(0);
      const int nslot = slot == 0 ? 2 : slot - 1;  // slot of tile t+2 = slot of tile t-1
      constexpr int NMF = 2 * FM * FN, GAP = NMF / (PW + 1);
#pragma unroll
      for (int q = 0; q < NMF; q++) {
        const int kk = q / (FM * FN), a = (q / FN) % FM, b = q % FN;
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][b], af[kk][a], acc[a][b], 0, 0, 0);
        if ((q + 1) % GAP == 0 && (q + 1) / GAP - 1 < PW) {
          __builtin_amdgcn_sched_barrier(0);
          piece(t + 2, nslot, (q + 1) / GAP - 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bar();
      slot = slot == 2 ? 0 : slot + 1;
    }
  } else {
    if (grp1) bar();  // group 1 runs one barrier behind group 0
    // per-phase DMA pieces of tile t+2 (this wave's PW pieces spread over the four phases, fewest next to the 10-read phase)
    constexpr int D0 = PW / 4, D1 = (PW + 2) / 4, D2 = (PW + 1) / 4, D3 = (PW + 3) / 4;
    static_assert(D0 + D1 + D2 + D3 == PW, "pieces");
    int slot = 0;
    h8 a0[2][FMh] = {}, a1[2][FMh] = {}, wa[2][FNa] = {}, wb[2][FNb > 0 ? FNb : 1] = {};
    for (int t = 0; t < nk; t++) {
      const char* s = smem + slot * TILE_BYTES;
      const int nslot = slot == 0 ? 2 : slot - 1;
      // ---- phase 0: quadrant (A-sub0, Wa) ----
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FNa; b++) wa[kk][b] = ldsr<MODE>(s + w_rd + b * 2048 + coff, wa[kk][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int a = 0; a < FMh; a++) a0[kk][a] = ldsr<MODE>(s + a_rd + a * 2048 + coff, a0[kk][a]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < D0; i++) piece(t + 2, nslot, i);
      bar();
      lgk0();
      if (MODE & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int a = 0; a < FMh; a++)
#pragma unroll
          for (int b = 0; b < FNa; b++) acc[a][b] = mma<MODE>(wa[kk][b], a0[kk][a], acc[a][b]);
      if (MODE & 2) __builtin_amdgcn_s_setprio(0);
      bar();
      // ---- phase 1: quadrant (A-sub0, Wb) ----
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FNb; b++) wb[kk][b] = ldsr<MODE>(s + w_rd + (FNa + b) * 2048 + coff, wb[kk][b]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < D1; i++) piece(t + 2, nslot, D0 + i);
      bar();
      lgk0();
      if (MODE & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int a = 0; a < FMh; a++)
#pragma unroll
          for (int b = 0; b < FNb; b++) acc[a][FNa + b] = mma<MODE>(wb[kk][b], a0[kk][a], acc[a][FNa + b]);
      if (MODE & 2) __builtin_amdgcn_s_setprio(0);
      bar();
      // ---- phase 2: quadrant (A-sub1, Wb) ----
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int a = 0; a < FMh; a++) a1[kk][a] = ldsr<MODE>(s + a_rd + (FMh + a) * 2048 + coff, a1[kk][a]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < D2; i++) piece(t + 2, nslot, D0 + D1 + i);
      bar();
      lgk0();
      if (MODE & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int a = 0; a < FMh; a++)
#pragma unroll
          for (int b = 0; b < FNb; b++)
            acc[FMh + a][FNa + b] = mma<MODE>(wb[kk][b], a1[kk][a], acc[FMh + a][FNa + b]);
      if (MODE & 2) __builtin_amdgcn_s_setprio(0);
      bar();
      // ---- phase 3: quadrant (A-sub1, Wa): no LDS reads; the K tile's one counted wait: tile t+1 has landed, tile t+2 stays in flight ----
#pragma unroll
      for (int i = 0; i < D3; i++) piece(t + 2, nslot, D0 + D1 + D2 + i);
      if (pw_hi) wait_vmcnt<PW>(); else wait_vmcnt<PW - 1>();
      bar();
      if (MODE & 2) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int a = 0; a < FMh; a++)
#pragma unroll
          for (int b = 0; b < FNa; b++)
            acc[FMh + a][b] = mma<MODE>(wa[kk][b], a1[kk][a], acc[FMh + a][b]);
      if (MODE & 2) __builtin_amdgcn_s_setprio(0);
      bar();
      slot = slot == 2 ? 0 : slot + 1;
    }
    if ((MODE & 1) && !grp1) bar();  // group 0 waits for group 1's last phase
  }
  wait_vmcnt<0>();
  // direct store: lane holds row m = a*16 + (lane & 15), columns b*16 + 4*(lane >> 4) + r
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

template <int WGM, int WGN, int FM, int FN, int MODE>
void run(const char* name, int M, int N, int K, half_t* A, half_t* W, half_t* C, float* Cref, int mstep) {
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16, LDS = 3 * (BM + BN) * 128;
  if (M % BM || N % BN || K % 64 || ((M / BM) * (N / BN)) % 8) { printf("%-44s %6dx%5dx%5d  (shape not tileable)\n", name, M, N, K); return; }
  auto fn = k8p<WGM, WGN, FM, FN, MODE>;
  (void)hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  const int tiles_n = N / BN, nt = (M / BM) * tiles_n;
  (void)hipMemset(C, 0, (size_t)M * N * 2);
  fn<<<nt, WGM * WGN * 64, LDS>>>(A, W, C, M, N, K, tiles_n);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("%-44s launch failed: %s\n", name, hipGetErrorString(e)); exit(1); }
  // check the sampled rows against the reference
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
    for (int i = 0; i < iters; i++) fn<<<nt, WGM * WGN * 64, LDS>>>(A, W, C, M, N, K, tiles_n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = fminf(best, ms / iters); tot += ms / iters;
  }
  const double fl = 2.0 * M * N * K;
  printf("%-44s %6dx%5dx%5d  %4d blocks  %8.2f us (best %8.2f)  %7.1f TF (best %7.1f)  rel_l2 %.2e %s\n", name, M, N, K, nt, tot / rounds * 1e3, best * 1e3,
         fl / (tot / rounds * 1e-3) / 1e12, fl / (best * 1e-3) / 1e12, rel, (MODE & 56) ? "(ablation)" : rel < 2e-3 ? "ok" : "WRONG");
}

int main(int argc, char** argv) {
  struct Shape { int M, N, K; };
  const Shape shapes[] = {{32768, 320, 2880}, {32768, 320, 5760}, {8192, 5120, 640}, {2048, 10240, 1280}, {8192, 640, 5760}, {16384, 2560, 2560}, {8192, 8192, 4096}};
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
    run<4, 2, 4, 5, 4>("256x160 lockstep (cfg-11 schedule)", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 0>("256x160 4-phase, no stagger, no setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 1>("256x160 4-phase + stagger", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 2>("256x160 4-phase + setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 5, 3>("256x160 4-phase + stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    run<4, 2, 4, 4, 4>("256x128 lockstep", M, N, K, A, W, C, Cref, mstep);
    if (s.N == 320 || s.K == 4096) {
      run<4, 2, 4, 5, 3 + 8>("  ablation: no DMA in the loop", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 16>("  ablation: no MFMA", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 32>("  ablation: no fragment reads", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 8 + 32>("  ablation: MFMA + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 16 + 32>("  ablation: DMA + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 8 + 16>("  ablation: fragment reads + barriers only", M, N, K, A, W, C, Cref, mstep);
      run<4, 2, 4, 5, 3 + 8 + 16 + 32>("  ablation: barriers only", M, N, K, A, W, C, Cref, mstep);
    }
    run<4, 2, 4, 4, 3>("256x128 4-phase + stagger + setprio", M, N, K, A, W, C, Cref, mstep);
    (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(Cref);
  }
  return 0;
}
