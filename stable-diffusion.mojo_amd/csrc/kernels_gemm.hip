// kernels_gemm.hip - fp16 MFMA GEMM and implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
// Replaces the reference's `Matrix.matmul` (helpers/utils.mojo:1549-1569), `Linear.forward`
// (:1954-1976) and `Conv2D.forward` (:1738-1811, incl. `Matrix.pad` :1383-1413 folded into the
// tap addressing and `Upsample` :1989-2010 folded into the source addressing).
//
// C[m][n] = epilogue( sum_k A[m][k] * W[n][k] )       ("NT": both operands K-contiguous)
//   dense : A[m][k] row-major (optionally the channel-concat of two tensors)
//   conv  : A row m = output pixel (b,oy,ox); k = (tap, cin) with the NHWC source read at
//           (oy*stride-pad+kh, ox*stride-pad+kw); out-of-image taps are zero-filled by the DMA's range check.
//
// Design (CDNA4, wave64) - DESIGN.md section 4.1 has the measurements behind each point:
//   * v_mfma_f32_16x16x32_f16, operands swapped (D = Wfrag x Afrag^T) so each lane ends up with ONE output row m and
//     4*FN CONSECUTIVE output columns n.  The column permutation that makes them consecutive is applied for free on
//     the SOURCE address of the W-tile load.
//   * Tiles staged by LDS-DMA through buffer descriptors (lds_dma.h): 16 B/lane straight into LDS, a 32-bit byte
//     offset per DMA instruction + the K position in the scalar offset (no address VALU), 128-B rows XOR-swizzled by
//     (row & 7) on the source side so every ds_read_b128 fragment read is bank-conflict-free.  LDS ring of NS slots,
//     counted s_waitcnt vmcnt, one raw s_barrier per K-tile.
//   * Pinned issue order in the K loop: all fragment reads, then the MFMAs with the next tile's DMA instructions
//     dropped one at a time into their shadow.
//   * Epilogue: residual tile prefetched before the DMA drain; per-column terms in the MFMA layout; per-wave LDS
//     transpose to row-major so loads/stores are 16 B per lane over whole row segments; optional GroupNorm
//     statistics (EPI_GNSTATS) for the consumer norm; split-K hand-off through the XCD-local L2 for the 16x16 level.
//   * XCD-aware block -> tile map: the 8 XCDs own an (8/xn) x xn grid of the tile space chosen by operand footprint.
//   * Round 3: optional loader waves (LW) that issue the block's whole LDS-DMA stream, and with eight compute waves the staggered
//     two-group schedule (tile configurations 51 / 53) - DESIGN.md 4.1 "what a K tile costs".
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <vector>

#include "common.h"
#include "lds_dma.h"

#define TSD_STR2(x) #x
#define TSD_STR(x) TSD_STR2(x)
#ifndef TSD_GEMM_LOOP_ALIGN
#define TSD_GEMM_LOOP_ALIGN 8
#endif
#ifndef TSD_GEMM_PIN
#define TSD_GEMM_PIN 1
#endif
#ifndef TSD_GEMM_PIPE
#define TSD_GEMM_PIPE 1
#endif
#ifndef TSD_GEMM_PIN_MAXNS
#define TSD_GEMM_PIN_MAXNS 4
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct GemmK {
  const half_t* A0; const half_t* A1; const half_t* Wt; const half_t* R; const half_t* zeros;
  const float* bias; const float* rowvec;
  void* C;
  long long sA, sW, sC, sR;
  int lda0, lda1, K0, ldw, ldr, ldc, rowvec_ld, rows_per_batch;
  int M, N, K;
  int Hs, Ws, Ho, Wo, Cin, stride, pad, ups;
  // conv3x3 with a fused 1x1 convolution of a second tensor (the residual block's skip path): K continues past the nine taps
  // with "taps" 9 / 10 = the centre pixel of A1 (Cin1 channels) / A2 (Cin2 channels, the second half of a channel concat),
  // weights Wt1[n][Cin1 + Cin2]
  const half_t* A2; const half_t* Wt1; int lda2, Cin1, Cin2, ldw1;
  int epi, tiles_n, xcd_n;
  unsigned w_kts;  // bytes from one K tile of W to the next (128: row-major)
  float out_scale;
  float* gn_part; int gn_cpg, gn_G, gn_hw, gn_nslab;  // EPI_GNSTATS
  half_t* vt; long long vt_sB; int vt_n0, vt_ld, vt_S;  // transposed tail: columns >= vt_n0 go to vt[b][n - vt_n0][s] (GemmArgs::Vt)
  float* sk_ws; int* sk_flags; int splitk; int sk_cfg; int sk_epoch;  // split-K (2/4/8 slices): fp32 partial tiles + one arrival flag per (slice, tile); host: tile cfg
#ifdef TSD_GEMM_TS
  unsigned long long* ts;  // per-block phase timestamps (experiment build only)
#endif
};

__device__ __forceinline__ float gelu_tanh_f(float x) {
  // 0.5 x (1 + tanh(u)) = x * sigmoid(2u) = x / (1 + 2^(-2 u log2 e)), u = sqrt(2/pi) (x + 0.044715 x^3)  (helpers/utils.mojo:1914).
  // One v_exp_f32 + one v_rcp_f32 (1 ulp) instead of expf + an IEEE division: the GEGLU epilogue evaluates this 80 times
  // per lane and tile, and the division sequence alone made it cost more than the K loop of the short-K GEMM it ends.
  const float c2 = -2.f * 0.7978845608028654f * 1.4426950408889634f;
  const float t = __builtin_amdgcn_exp2f(c2 * (x + 0.044715f * x * x * x));
  return x * __builtin_amdgcn_rcpf(1.f + t);
}

#ifdef TSD_GEMM_TS
static unsigned long long* g_ts = nullptr;
#define TS_MARK(i) do { if (p.ts && threadIdx.x == 0) { p.ts[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (i)] = __builtin_amdgcn_s_memtime(); if ((i) == 0 || (i) == 4) p.ts[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + ((i) == 0 ? 5 : 6)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define TS_MARK(i) do { } while (0)
#endif

// N instructions may stay in flight: OTHER plain 16-byte loads on top of N - OTHER LDS-DMA pieces of younger tiles, PPT pieces per tile
// and wave (lds_dma.h: the accounting travels into the assembly, tools/isa_lint.py checks it against what hipcc emitted)
template <int N, int PPT = 0, int OTHER = 0, int PPT2 = 0>
__device__ __forceinline__ void wait_vmcnt() { wait_vm_counted<N - OTHER, OTHER, PPT, PPT2>(); }

// NS = LDS ring depth.  NS == 2: two blocks per CU hide the DMA latency by TLP (big M).  NS >= 3: one block per
// CU with NS-1 K-tiles of DMA in flight behind counted s_waitcnt vmcnt (few-tile problems: M = 2048 level of the
// UNet, where only 256 tiles exist and every iteration would otherwise expose a full HBM/L2 round trip).
// PP ("ping-pong", 8 waves, NS == 3): waves 0-3 and 4-7 share the four SIMDs pairwise and run half an iteration apart -
// while one group issues its 2*FM*FN MFMAs of K-tile t the other group reads its fragments of the next tile from LDS
// and issues DMA, so each SIMD's matrix pipe always has a wave feeding it.  Two barriers per K-tile separate the
// phases; K-tiles are DMA'd three ahead into a 3-slot ring behind counted s_waitcnt vmcnt.
// HX ("halo in x", conv3x3 stride 1, 128-row tiles, 2-slot ring): K is visited as (kh, channel chunk, kw) and the A operand
// of the three kw taps of a (kh, chunk) group is ONE staged tile of the tile's pixels plus a left / right halo pixel per
// image row (BM + 2 rows per image row segment); the kw taps read it at a row offset of kw.  The input pixels cross the
// L2 -> CU path three times per channel chunk instead of nine: 71 -> 100 flop per L2 byte for the 128x160 tile, and that
// path is what bounds these convs.  The W ring keeps its 2 slots; the A tiles live in two buffers of
// their own (a group's tile must outlive three ring rotations).
// LW ("loader waves", 0 or 4): LW extra waves per block - one per SIMD - issue every LDS-DMA instruction of the block and do
// nothing else; the NW compute waves only read fragments and multiply.  An LDS-DMA instruction costs its issuing wave 60-185
// cycles (MI355X_MICROARCH.md, per-instruction constants), during which that wave issues no MFMA: with one compute wave per SIMD
// (the one-block-per-CU configurations) the matrix pipe idles for every one of the 4-7 DMA instructions per K tile.  Measured in
// isolation (scripts/micro/gemm_ws.hip + profiles/r03_micro_gemm_ws.txt): 822 -> 957 TF with loaders, 1137 TF with staggered groups.
// The loaders run the same ring protocol (counted vmcnt, one barrier per K tile) and end before the epilogue.
template <int WGM, int WGN, int FM, int FN, bool CONV, int NS, bool PP = false, int CV = 0, int LW = 0>
__global__ __launch_bounds__((WGM * WGN + LW) * 64, LW ? (WGM * WGN + LW) / 4 : ((NS <= 2 || WGM * WGN > 4) && FM * FN <= 20 ? 2 : 1)) void gemm_kernel(const GemmK p) {
  constexpr int NW = WGM * WGN;
  constexpr int NI = LW ? LW : NW;  // waves that issue DMA
  constexpr bool HX = CV == 1;  // conv variant: 1 = halo-x K order, 2 = fused 1x1 skip source (K runs on past the nine taps)
  constexpr bool SK = CV == 2;
  static_assert(CV == 0 || CONV, "conv variants");
  static_assert(LW == 0 || (LW == 4 && !PP && !HX), "loader waves: one per SIMD, plain ring schedules only");
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16;
  constexpr int BMw = FM * 16, BNw = FN * 16;
  constexpr int A_INSTR = BM / 8, W_INSTR = BN / 8;  // 1-KiB wave-instructions per tile
  constexpr int A_PW = (A_INSTR + NI - 1) / NI, W_PW = (W_INSTR + NI - 1) / NI;
  constexpr int TILE_BYTES = (BM + BN) * 128;
  static_assert(!HX || (CONV && NS == 2 && !PP && NW == 4 && BM == 128 && BMw == 64), "halo-x: 128-row conv tiles, 2x2 waves, 2-slot ring");
  constexpr int WB = BN * 128;                              // HX: bytes of a ring slot (W tile only)
  constexpr int AH_INSTR = BM / 8 + 1, AB = AH_INSTR * 1024;  // HX: halo tile = BM + 2*nr rows (<= BM + 8), two buffers after the ring
  constexpr int A_PWX = HX ? (AH_INSTR + NI - 1) / NI : A_PW;  // A DMA instructions per wave and staged tile
  extern __shared__ __attribute__((aligned(16))) char smem[];

  TS_MARK(0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const bool loader = LW && wave >= NW;      // wave-uniform
  const bool issuer = LW ? loader : true;    // this wave owns DMA pieces
  const int iw = LW ? wave - NW : wave;      // its index among the issuing waves
  __builtin_assume(!issuer || (iw >= 0 && iw < NI));  // (the per-piece `j < W_INSTR` tests fold to the one that can fail)

  // XCD-aware bijective remap (block b runs on XCD b%8; give each XCD a contiguous tile range).
  // The 8 XCDs form an (8/xcd_n) x xcd_n grid over the tile space: an XCD owns a contiguous block of tile rows AND of
  // tile columns, so through its private L2 it pulls 1/xcd_n of W and xcd_n/8 of A.  The host picks xcd_n to minimise
  // the fabric traffic W_bytes * (8/xcd_n) + A_bytes * xcd_n (weight-heavy M = 2048 problems want xcd_n = 8: with
  // row-only ownership every XCD re-fetched the whole 29 MB conv weight).  xcd_n = 1 is the row-only bijective remap.
  // Split-K (p.splitk = S in {2, 4, 8}, few-tile problems with a long K): the grid holds every tile S times.  Slice s
  // owns the s-th contiguous share of K; the blocks of slices S-1 .. 1 come first in the grid and hand their fp32
  // partial tile to the slice-0 block (the last `ntile` ids; same XCD as its producers, so the same L2), which adds
  // the partials in slice order and runs the epilogue.
  const int S = p.splitk;
  const int ntile = S > 1 ? (int)gridDim.x / S : (int)gridDim.x;
  const int ks = S > 1 ? S - 1 - (int)blockIdx.x / ntile : 0;
  int bid = S > 1 ? (int)blockIdx.x - (S - 1 - ks) * ntile : (int)blockIdx.x;
  int tm, tn;
  if (p.xcd_n > 1) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int xi = xcd / p.xcd_n, xj = xcd - xi * p.xcd_n;
    const int tnx = p.tiles_n / p.xcd_n, tmx = (ntile / p.tiles_n) / (8 / p.xcd_n);
    const int ltm = idx / tnx, ltn = idx - ltm * tnx;
    tm = xi * tmx + ltm;
    tn = xj * tnx + ltn;
  } else {
    const int nwg = ntile, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    tm = bid / p.tiles_n;
    tn = bid - tm * p.tiles_n;
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk_all = p.K >> 6;
  const int kt0 = S > 1 ? (ks * nk_all) / S : 0;                      // first K-tile of this block
  const int nk = S > 1 ? ((ks + 1) * nk_all) / S - kt0 : nk_all;      // and how many it owns
  const int bz = blockIdx.y;
  const half_t* A0 = p.A0 + (long long)bz * p.sA;
  const half_t* A1 = p.A1 ? p.A1 + (long long)bz * p.sA : nullptr;
  const half_t* Wt = p.Wt + (long long)bz * p.sW;
  // Fused-skip sources as DIFFERENCES to the main ones, picked by mask arithmetic below.  Written as `c ? A0 : c2 ? A1 : A2`
  // inside the staging lambdas, hipcc turns the three by-reference captures into one load at a run-time offset into the closure
  // object; the closure then stays in memory, and with it every variable it refers to - 500-700 B of scratch per lane.
  const long long dA1 = SK ? (long long)((uintptr_t)p.A1 - (uintptr_t)p.A0) : 0, dA2 = SK ? (long long)((uintptr_t)p.A2 - (uintptr_t)p.A0) : 0;
  const long long dW1 = SK ? (long long)((uintptr_t)p.Wt1 - (uintptr_t)p.Wt) : 0;
  const unsigned k_lda1 = (unsigned)p.lda1, k_dlda2 = (unsigned)(p.lda2 - p.lda1), k_ldw1 = (unsigned)p.ldw1, k_cin1 = (unsigned)p.Cin1;

  const int lrow = lane >> 3;
  const int cch = (lane & 7) ^ lrow;  // logical 16-B chunk this lane fetches (source-side swizzle)

  // ---- per-thread source offsets ---------------------------------------------------------
  // 32-bit BYTE offsets from the (wave-uniform) descriptor bases; the K position goes in the scalar offset, so a
  // DMA instruction costs no address VALU at all (every operand slice on the path is < 2 GiB, checked by the host).
  unsigned a_off[A_PWX];   // dense: row offset in the first source
  unsigned a_off1[A_PWX];  // dense: row offset in the second concat source ; conv: current tap offset (PAD_OFF = zero tap)
  unsigned a_pk[A_PWX];    // conv: (oy*stride-pad+1) | (ox*stride-pad+1) << 11 | b << 22   (b = 1023: row beyond M)
  // HX geometry (wave-uniform): the tile is nr = 128 / seg image-row segments of seg = min(Wo, 128) pixels
  const int hx_seg = HX ? (p.Wo < 128 ? p.Wo : 128) : 1;
  if (issuer) {
#pragma unroll
  for (int i = 0; i < A_PWX; i++) {
    const int row = (iw + i * NI) * 8 + lrow;
    int m = m0 + row;
    const bool ok = m < p.M;
    if (!ok) m = p.M - 1;
    if constexpr (HX) {
      // halo tile row R: image-row segment ir, column c in -1 .. seg (the two ends are the halo pixels)
      const int ir = row / (hx_seg + 2), c = row - ir * (hx_seg + 2) - 1;
      const bool valid = row < BM + 2 * (BM / hx_seg);
      const int hw = p.Ho * p.Wo, b = m0 / hw, rem = m0 - b * hw;
      const int oy = rem / p.Wo + ir, ox = rem - (rem / p.Wo) * p.Wo + c;
      a_pk[i] = (unsigned)(oy + 1) | (unsigned)(ox + 1) << 11 | (valid ? (unsigned)b : 1023u) << 22;  // input row of tap kh: oy + kh - 1
      a_off[i] = 0;
      a_off1[i] = PAD_OFF;
    } else if constexpr (!CONV) {
      a_off[i] = ((unsigned)m * (unsigned)p.lda0 + cch * 8) * 2;
      a_off1[i] = ((unsigned)m * (unsigned)p.lda1 + cch * 8) * 2;
      a_pk[i] = 0;
    } else {
      const int hw = p.Ho * p.Wo;
      const int b = m / hw, rem = m - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      a_pk[i] = (unsigned)(oy * p.stride - p.pad + 1) | (unsigned)(ox * p.stride - p.pad + 1) << 11 | (ok ? (unsigned)b : 1023u) << 22;
      a_off[i] = 0;
      a_off1[i] = PAD_OFF;
    }
  }
  }
  unsigned w_off[W_PW];
  if (issuer) {
#pragma unroll
  for (int i = 0; i < W_PW; i++) {
    const int rho = (iw + i * NI) * 8 + lrow;  // LDS row of the W tile
    const int wq = rho / BNw, rr = rho - wq * BNw, fn = rr >> 4, ii = rr & 15;
    int n = n0 + wq * BNw + (ii >> 2) * (4 * FN) + fn * 4 + (ii & 3);  // column permutation
    if (n >= p.N) n = p.N - 1;
    w_off[i] = ((unsigned)n * (unsigned)p.ldw + cch * 8) * 2;
  }
  }

  // conv: running (tap, channel-chunk) of the NEXT tile to stage
  const int cpt = CONV ? (p.Cin >> 6) : 1;
  const int cpt1 = SK ? (p.Cin1 >> 6) : 0, cpt2 = SK ? (p.Cin2 >> 6) : 0;  // fused 1x1 skip sources
  const int ntaps = SK ? 9 + (cpt1 > 0) + (cpt2 > 0) : 9;
  int st_tap = kt0 / cpt, st_cc = kt0 - (kt0 / cpt) * cpt;
  if (SK && kt0 >= 9 * cpt) {  // a split-K slice that starts inside the skip segment
    st_tap = 9; st_cc = kt0 - 9 * cpt;
    if (st_cc >= cpt1) { st_tap = 10; st_cc -= cpt1; }
  }
  int st_kw = 0, st_g = 0;  // HX: (st_tap = kh, st_cc = chunk, st_kw) of the next tile to stage; st_g = its (kh, chunk) group number
  auto conv_tap_ptrs = [&](int tap) {
    if constexpr (HX) {  // tap = kh: source offsets of the halo tile rows
#pragma unroll
      for (int i = 0; i < A_PWX; i++) {
        const int iy = (int)(a_pk[i] & 2047u) - 1 + tap - 1, ix = (int)((a_pk[i] >> 11) & 2047u) - 1;
        const unsigned b = a_pk[i] >> 22;
        const bool ok = b != 1023u && (unsigned)iy < (unsigned)p.Hs && (unsigned)ix < (unsigned)p.Ws;
        a_off1[i] = ok ? ((unsigned)(((int)b * p.Hs + iy) * p.Ws + ix) * (unsigned)p.lda0 + cch * 8) * 2 : PAD_OFF;
      }
      return;
    }
    // One loop for the nine taps and for the fused 1x1 skip "taps" 9 / 10 (the output pixel itself in A1 / A2; stride 1, same resolution:
    // checked by the host), the variants picked by selects.  Written as two loops under `if (tap >= 9)` hipcc merged their tails and
    // indexed a_off1[] through a register: the three 16-byte offset tables of the N = 128 fused-skip kernels moved to scratch (48 B per
    // lane, csrc/isa_contract.json) and dispatch() had to avoid the 128x128 tile for the VAE's 256 -> 128 residual block.
    const bool skip = SK && tap >= 9;  // wave-uniform
    const int kh = tap / 3, kw = tap - kh * 3;
    const int dy = skip ? p.pad : kh, dx = skip ? p.pad : kw;
    const bool ups = !skip && p.ups;
    const int Heff = ups ? 2 * p.Hs : p.Hs, Weff = ups ? 2 * p.Ws : p.Ws;
    const unsigned ld = skip ? k_lda1 + (k_dlda2 & (tap == 9 ? 0u : ~0u)) : (unsigned)p.lda0;
#pragma unroll
    for (int i = 0; i < A_PW; i++) {
      const int iy = (int)(a_pk[i] & 2047u) - 1 + dy, ix = (int)((a_pk[i] >> 11) & 2047u) - 1 + dx;
      const unsigned b = a_pk[i] >> 22;
      const bool ok = b != 1023u && (skip || ((unsigned)iy < (unsigned)Heff && (unsigned)ix < (unsigned)Weff));
      const int sy = ups ? iy >> 1 : iy, sx = ups ? ix >> 1 : ix;
      a_off1[i] = ok ? ((unsigned)(((int)b * p.Hs + sy) * p.Ws + sx) * ld + cch * 8) * 2 : PAD_OFF;
    }
    if (skip) {
#pragma unroll
      for (int i = 0; i < W_PW; i++) {  // the skip weights have their own row pitch
        const int rho = (iw + i * NI) * 8 + lrow;
        const int wq = rho / BNw, rr = rho - wq * BNw, fn = rr >> 4, ii = rr & 15;
        int n = n0 + wq * BNw + (ii >> 2) * (4 * FN) + fn * 4 + (ii & 3);
        if (n >= p.N) n = p.N - 1;
        w_off[i] = ((unsigned)n * k_ldw1 + cch * 8) * 2;
      }
    }
  };
  if constexpr (CONV) { if (issuer) conv_tap_ptrs(st_tap); }

  // One K-tile = A_PW + W_PW DMA instructions per wave.  A TileSrc holds the wave-uniform part of their addresses;
  // stage_piece() issues the i-th instruction so the pinned schedule can drop them one at a time into the shadow of
  // the MFMAs; stage_advance() steps the conv (tap, chunk) cursor.  `live` = false builds descriptors with zero
  // records: every lane is out of range and the DMA writes zeros, which lets the loop tail keep the same
  // straight-line body instead of branching around the DMA.
  struct TileSrc { rsrc_t ra, rw; unsigned a_soff, w_soff; bool second; bool a_on; int abuf; };
  auto tile_src = [&](int kt, bool live) {
    TileSrc t;
    const int nrec = live ? 0x7ffffff0 : 0;
    const int k0 = (kt0 + kt) * 64;
    t.second = !CONV && k0 >= p.K0;  // wave-uniform: second concat source
    const bool skip = SK && st_tap >= 9;  // wave-uniform
    if constexpr (CONV) {
      if constexpr (SK) {
        const long long da = (dA1 & (st_tap == 9 ? -1LL : 0LL)) + (dA2 & (st_tap >= 10 ? -1LL : 0LL));
        t.ra = __builtin_amdgcn_make_buffer_rsrc((half_t*)((uintptr_t)A0 + (uintptr_t)da), 0, nrec, 0x00020000);
      } else {
        t.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(A0), 0, nrec, 0x00020000);
      }
      t.a_soff = st_cc * 128;
    } else {
      t.ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(t.second ? A1 : A0), 0, nrec, 0x00020000);
      t.a_soff = (t.second ? k0 - p.K0 : k0) * 2;
    }
    if constexpr (SK) {
      t.rw = __builtin_amdgcn_make_buffer_rsrc((half_t*)((uintptr_t)Wt + (uintptr_t)(dW1 & (skip ? -1LL : 0LL))), 0, nrec, 0x00020000);
      t.w_soff = skip ? ((st_tap == 9 ? 0u : k_cin1) + (unsigned)st_cc * 64) * 2 : (unsigned)(kt0 + kt) * p.w_kts;
    } else {
      t.rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(Wt), 0, nrec, 0x00020000);
      t.w_soff = (unsigned)(kt0 + kt) * p.w_kts;
    }
    t.a_on = true; t.abuf = 0;
    if constexpr (HX) {
      t.w_soff = (unsigned)((st_tap * 3 + st_kw) * p.Cin + st_cc * 64) * 2;
      t.a_on = st_kw == 0;  // the group's halo tile travels with its first tap
      t.abuf = st_g & 1;
    }
    return t;
  };
  auto stage_piece = [&](const TileSrc& t, int buf, int i) {
    char* sA = HX ? smem + NS * WB + t.abuf * AB : smem + buf * TILE_BYTES;
    char* sW = HX ? smem + buf * WB : sA + BM * 128;
    if (i < A_PWX) {
      const int j = iw + i * NI;
      if constexpr (HX) {
        if (t.a_on && j < AH_INSTR) blds16(t.ra, a_off1[i], t.a_soff, sA + j * 1024);
      } else if (A_INSTR % NI == 0 || j < A_INSTR) {
        if constexpr (!CONV) blds16(t.ra, t.second ? a_off1[i] : a_off[i], t.a_soff, sA + j * 1024);
        else blds16(t.ra, a_off1[i], t.a_soff, sA + j * 1024);
      }
    } else {
      const int ii = i - A_PWX;
      const int j = iw + ii * NI;
      if (W_INSTR % NI == 0 || j < W_INSTR) blds16(t.rw, w_off[ii], t.w_soff, sW + j * 1024);
    }
  };
  auto stage_advance = [&]() {
    if constexpr (HX) {
      if (++st_kw == 3) {
        st_kw = 0;
        ++st_g;
        if (++st_cc == cpt) {
          st_cc = 0;
          ++st_tap;
          if (st_tap < 3) conv_tap_ptrs(st_tap);
        }
      }
    } else if constexpr (CONV) {
      const int lim = SK ? cpt + ((cpt1 - cpt) & (st_tap == 9 ? -1 : 0)) + ((cpt2 - cpt) & (st_tap >= 10 ? -1 : 0)) : cpt;
      if (++st_cc == lim) {
        st_cc = 0;
        ++st_tap;
        if (st_tap < ntaps) conv_tap_ptrs(st_tap);
      }
    }
  };
  auto stage = [&](int kt, int buf) {
    const TileSrc t = tile_src(kt, true);
#pragma unroll
    for (int i = 0; i < A_PWX + W_PW; i++) stage_piece(t, buf, i);
    stage_advance();
  };

  f4 acc[FM][FN];
#pragma unroll
  for (int a = 0; a < FM; a++)
#pragma unroll
    for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets: row R = base + (lane&15); chunk c = kk*4 + (lane>>4); phys = c ^ (R&7)
  const int rsel = lane & 15, key = lane & 7, cq = lane >> 4;
  const int a_rd = (wm * BMw + rsel) * 128, w_rd = (wn * BNw + rsel) * 128;

  auto compute = [&](int buf) {
    const char* sA = smem + buf * TILE_BYTES;
    const char* sW = sA + BM * 128;
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const int coff = ((kk * 4 + cq) ^ key) << 4;
      h8 af[FM], wf[FN];
#pragma unroll
      for (int a = 0; a < FM; a++) af[a] = *(const h8*)(sA + a_rd + a * 2048 + coff);
#pragma unroll
      for (int b = 0; b < FN; b++) wf[b] = *(const h8*)(sW + w_rd + b * 2048 + coff);
#pragma unroll
      for (int a = 0; a < FM; a++)
#pragma unroll
        for (int b = 0; b < FN; b++)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[b], af[a], acc[a][b], 0, 0, 0);
    }
  };

  // whole-K-tile fragment set in registers (pinned-order and ping-pong schedules)
  constexpr bool PIN = !PP && (NS <= TSD_GEMM_PIN_MAXNS) && (FM * FN * 4 + 8 * (FM + FN) + 56 <= 256) && (TSD_GEMM_PIN != 0);
  h8 af[2][(PIN || PP) ? FM : 1], wf[2][(PIN || PP) ? FN : 1];
  static_assert(!HX || PIN, "halo-x runs the pinned schedule");
  int c_kw = 0, c_g = 0;  // HX: kw and group number of the tile being multiplied
  const int hx_base = HX ? wm * BMw + 2 * ((wm * BMw) / hx_seg) : 0;  // halo-tile row of this wave's first pixel at kw = 0 (its left halo)
  auto read_frags = [&](int buf) {
    if constexpr (HX) {
      const char* sA = smem + NS * WB + (c_g & 1) * AB;
      const char* sW = smem + buf * WB;
      const int hr = hx_base + c_kw + rsel, hkey = hr & 7;  // halo-tile row of fragment 0, its swizzle key (fragments are 16 rows apart)
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4, coffa = ((kk * 4 + cq) ^ hkey) << 4;
#pragma unroll
        for (int b = 0; b < FN; b++) wf[kk][b] = *(const h8*)(sW + w_rd + b * 2048 + coff);
#pragma unroll
        for (int a = 0; a < FM; a++) af[kk][a] = *(const h8*)(sA + hr * 128 + a * 2048 + coffa);
      }
      if (++c_kw == 3) { c_kw = 0; ++c_g; }
    } else if constexpr (PIN || PP) {
      const char* sA = smem + buf * TILE_BYTES;
      const char* sW = sA + BM * 128;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FN; b++) wf[kk][b] = *(const h8*)(sW + w_rd + b * 2048 + coff);
#pragma unroll
        for (int a = 0; a < FM; a++) af[kk][a] = *(const h8*)(sA + a_rd + a * 2048 + coff);
      }
    }
  };
  auto mfma_all = [&]() {
    if constexpr (PIN || PP) {
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int a = 0; a < FM; a++)
#pragma unroll
          for (int b = 0; b < FN; b++)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][b], af[kk][a], acc[a][b], 0, 0, 0);
    }
  };

  // DMA instructions per stage per wave: waves below the remainder issue one more (wave-uniform)
  constexpr int LPS_HI = A_PW + W_PW;
  constexpr int LPS_LO = (A_INSTR % NI ? A_PW - 1 : A_PW) + (W_INSTR % NI ? W_PW - 1 : W_PW);
  static_assert(NS == 2 || A_INSTR % NI == 0, "A tile loads must divide evenly among the waves");
  const bool lps_hi = (W_INSTR % NI == 0) || iw < (W_INSTR % NI);
  // wait until at most `ahead` of this wave's most recent K tiles are still in flight.  EXACTLY ONE of the alternatives runs (an if / else
  // chain that ends in an unconditional else): the markers say so to tools/isa_lint.py, which cannot see it in hipcc's structurised
  // control flow (two independent skips around two waits) and would otherwise find a path around every wait
  constexpr int LPS_ALT = LPS_HI != LPS_LO ? LPS_LO : 0, LPS_ALT2 = LPS_HI != LPS_LO ? LPS_HI : 0;  // the other waves' piece count (0: all alike)
  auto wait_ring = [&](int ahead) {
    wait_alt_begin();
    if (NS >= 6 && ahead >= 4) { if (lps_hi) wait_vmcnt<4 * LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<4 * LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
    else if (NS >= 5 && ahead == 3) { if (lps_hi) wait_vmcnt<3 * LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<3 * LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
    else if (NS >= 4 && ahead == 2) { if (lps_hi) wait_vmcnt<2 * LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<2 * LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
    else if (NS >= 3 && ahead == 1) { if (lps_hi) wait_vmcnt<LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
    else wait_vmcnt<0>();
    wait_alt_end();
  };
  if constexpr (LW > 0) {
    if (loader) {
      // Loader wave: the block's whole DMA stream, same ring protocol as the compute waves' loops below - before barrier kt
      // tile kt has landed (counted vmcnt leaves the younger tiles in flight); after it every compute wave has retired its
      // reads of tile kt-1 (lgkmcnt(0) in front of its barrier), so that slot is refilled with tile kt+NS-1.  No dead tail
      // DMAs: nothing of this wave is in flight when it leaves, the epilogue's LDS staging starts behind barrier nk-1.
#pragma unroll
      for (int s = 0; s < NS - 1; s++)
        if (s < nk) stage(s, s);
      int nxt = NS - 1;
      if constexpr (NW == 8) {
        // staggered schedule (see the compute side below): four barriers per K tile, aligned with compute group 0; the wave's
        // pieces of tile kt+2 are spread over the four intervals, tile kt+1 has landed before the fourth barrier
        static_assert(NW != 8 || NS == 3, "staggered schedule: 3-slot ring");
        wait_alt_begin();
        if (nk > 1) wait_vmcnt<LPS_HI, LPS_HI>(); else wait_vmcnt<0>();  // tile 0 has landed (LPS_HI == LPS_LO: pieces divide evenly over 4 loaders)
        wait_alt_end();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        constexpr int NP = A_PW + W_PW;
        static_assert(NW != 8 || LPS_HI == LPS_LO, "loader pieces must divide evenly");
        // two loops (round 5) instead of `live = kt + 2 < nk` inside one: the tile two ahead exists in every trip of the first, in none
        // of the second - no branch around the DMA pieces, and a listing in which every path between two waits issues whole tiles
        int kt = 0;
        for (; kt + 2 < nk; kt++) {
          tsd_jitter();
          const TileSrc t = tile_src(kt + 2, true);
#pragma unroll
          for (int q = 0; q < 4; q++) {
#pragma unroll
            for (int i = q * NP / 4; i < (q + 1) * NP / 4; i++) stage_piece(t, nxt, i);
            if (q == 3) { stage_advance(); wait_vmcnt<NP, NP>(); }   // tile kt+1 has landed, tile kt+2 stays in flight
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          }
          nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
        for (; kt < nk; kt++) {   // the last two K tiles: nothing left to fetch
          tsd_jitter();
#pragma unroll
          for (int q = 0; q < 4; q++) {
            if (q == 3) wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
          }
        }
        __builtin_amdgcn_s_barrier();  // group 1's last phase
        return;
      }
      for (int kt = 0; kt < nk; kt++) {
        tsd_jitter();
        const int ahead = min(NS - 2, nk - 1 - kt);
        wait_ring(ahead);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + NS - 1 < nk) stage(kt + NS - 1, nxt);
        nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
      }
      return;
    }
  }
  if constexpr (LW > 0 && NW == 8) {
    // Staggered, wave-specialised schedule (scripts/micro/gemm_ws.hip: 1.14 PF where the lockstep loop reaches 0.82-0.89): the
    // eight compute waves form two groups that run one barrier apart, so on every SIMD one wave multiplies a 32-deep k-step
    // (FM x FN MFMAs from 9 fragment registers) while its partner reads the next k-step's fragments; the four loader waves
    // issue all DMA.  Same products in the same order as every other configuration: bitwise-identical results.
    const bool grp1 = wave >= NW / 2;
    __builtin_amdgcn_s_barrier();  // tile 0 has landed (loaders)
    asm volatile("" ::: "memory");
    if (grp1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    int cur = 0;
    for (int kt = 0; kt < nk; kt++) {
      tsd_jitter();
      const char* sA = smem + cur * TILE_BYTES;
      const char* sW = sA + BM * 128;
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + cq) ^ key) << 4;
        h8 afk[FM], wfk[FN];
#pragma unroll
        for (int b = 0; b < FN; b++) wfk[b] = *(const h8*)(sW + w_rd + b * 2048 + coff);
#pragma unroll
        for (int a = 0; a < FM; a++) afk[a] = *(const h8*)(sA + a_rd + a * 2048 + coff);
        __builtin_amdgcn_sched_barrier(0);
        // the tile's last reads retire BEFORE the barrier: the loaders refill its slot right after it
        if (kk == 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kk == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int a = 0; a < FM; a++)
#pragma unroll
          for (int b = 0; b < FN; b++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wfk[b], afk[a], acc[a][b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
      }
      cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
    if (!grp1) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
  } else if constexpr (!PP) {
    if constexpr (LW == 0) {
  #pragma unroll
    for (int s = 0; s < NS - 1; s++)
      if (s < nk) stage(s, s);
    }
    int cur = 0, nxt = NS - 1;  // ring slots of tile kt and tile kt+NS-1
    TS_MARK(1);
    // the K loop starts on a 256-byte boundary (padding = s_nop, executed once): +0.45 % on the step, measured
    asm volatile(".p2align " TSD_STR(TSD_GEMM_LOOP_ALIGN));
    // PIPE (one 4-wave block per CU, NS >= 3: nobody else on the SIMD hides LDS latency): software pipeline over the
    // K-tiles with two fragment register sets - the 2*(FM+FN) fragment reads of tile kt are issued one at a time between
    // the MFMAs of tile kt-1 instead of as a burst right after the barrier (72 ds_read_b128 from the four waves hold the
    // LDS, and every wave's issue, for ~300 cycles per K-tile).  Same products in the same order: results are bitwise
    // identical to the other schedules.
    // with loader waves the block runs two waves per SIMD (256 registers each): the second fragment set fits only for the 32-row wave tiles
    constexpr bool PIPE = PIN && (NS >= 3 || FM * FN <= 10) && NW == 4 && (TSD_GEMM_PIPE != 0) && (LW == 0 || FM * FN <= 10);
    if constexpr (PIPE) {
      h8 afA[2][FM], wfA[2][FN], afB[2][FM], wfB[2][FN];
      auto step = [&](int kt, h8 (&naf)[2][FM], h8 (&nwf)[2][FN], const h8 (&paf)[2][FM], const h8 (&pwf)[2][FN],
                      bool have_prev, bool have_next) {
        const char *sA = smem, *sW = smem;
        TileSrc t;
        if (have_next) {
          tsd_jitter();
          if constexpr (LW == 0) {
          const int ahead = min(NS - 2, nk - 1 - kt);
          wait_ring(ahead);
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile kt-1 are done: its slot may be refilled
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
          sA = smem + cur * TILE_BYTES;
          sW = sA + BM * 128;
          if constexpr (LW == 0) t = tile_src(kt + NS - 1, kt + NS - 1 < nk);
        }
        constexpr int NP = LW ? 0 : A_PW + W_PW, NM = 2 * FM * FN, NR = 2 * (FM + FN), GAP = NM / (NP + 1) > 0 ? NM / (NP + 1) : 1;
        auto read_one = [&](int i) {  // fragment i of tile kt: per k-half the FM A fragments, then the FN W fragments
          const int kk = i / (FM + FN), j = i - kk * (FM + FN);
          const int coff = ((kk * 4 + cq) ^ key) << 4;
          if (j < FM) naf[kk][j] = *(const h8*)(sA + a_rd + j * 2048 + coff);
          else nwf[kk][j - FM] = *(const h8*)(sW + w_rd + (j - FM) * 2048 + coff);
        };
        if (!have_prev) {
#pragma unroll
          for (int i = 0; i < NR; i++) read_one(i);
#pragma unroll
          for (int i = 0; i < NP; i++) stage_piece(t, nxt, i);
        } else {
#pragma unroll
          for (int q = 0; q < NM; q++) {
            const int kk = q / (FM * FN), a = (q / FN) % FM, b = q % FN;
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[kk][b], paf[kk][a], acc[a][b], 0, 0, 0);
            if (have_next) {
              __builtin_amdgcn_sched_barrier(0);
              if (q < NR) read_one(q);
              if ((q + 1) % GAP == 0 && (q + 1) / GAP - 1 < NP) stage_piece(t, nxt, (q + 1) / GAP - 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          if (have_next) {
#pragma unroll
            for (int i = NM; i < NR; i++) read_one(i);  // thin tiles: more fragments than MFMAs
#pragma unroll
            for (int i = NM / GAP; i < NP; i++) stage_piece(t, nxt, i);
          }
        }
        if (have_next) {
          if constexpr (LW == 0) stage_advance();
          cur = (cur + 1 == NS) ? 0 : cur + 1;
          nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
        }
      };
      step(0, afA, wfA, afB, wfB, false, true);
      int kt = 1;
      for (; kt + 1 < nk; kt += 2) {
        step(kt, afB, wfB, afA, wfA, true, true);
        step(kt + 1, afA, wfA, afB, wfB, true, true);
      }
      if (kt < nk) {
        step(kt, afB, wfB, afA, wfA, true, true);
        step(nk, afA, wfA, afB, wfB, true, false);
      } else {
        step(nk, afB, wfB, afA, wfA, true, false);
      }
    } else
    for (int kt = 0; kt < nk; kt++) {
      tsd_jitter();
      // tiles issued beyond kt so far: min(NS-2, nk-1-kt); wait until tile kt has landed, keep the rest in flight
      if constexpr (LW == 0) {
      const int ahead = min(NS - 2, nk - 1 - kt);
      wait_ring(ahead);
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads of tile kt-1 retired before the loaders may refill its slot
      }
      __builtin_amdgcn_s_barrier();  // every wave's share of tile kt is in LDS; slot of tile kt-1 is free
      asm volatile("" ::: "memory");
      if constexpr (PIN) {
        // Pinned issue order: all 2*(FM+FN) fragment reads first, then the MFMAs in fragment-arrival order behind the
        // compiler's counted lgkmcnt waits, with the next tile's DMA instructions dropped one at a time into the
        // MFMA shadow.  hipcc's own schedule front-loads the DMA address math and sinks some reads between MFMA
        // groups behind lgkmcnt(0), exposing 3-4 LDS round trips per K-tile.
        read_frags(cur);
        __builtin_amdgcn_sched_barrier(0);
        TileSrc t;
        if constexpr (LW == 0) t = tile_src(kt + NS - 1, kt + NS - 1 < nk);
        constexpr int NP = LW ? 0 : A_PWX + W_PW, NM = 2 * FM * FN, GAP = NM / (NP + 1) > 0 ? NM / (NP + 1) : 1;
#pragma unroll
        for (int q = 0; q < NM; q++) {
          const int kk = q / (FM * FN), a = (q / FN) % FM, b = q % FN;
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][b], af[kk][a], acc[a][b], 0, 0, 0);
          if ((q + 1) % GAP == 0 && (q + 1) / GAP - 1 < NP) {
            __builtin_amdgcn_sched_barrier(0);
            stage_piece(t, nxt, (q + 1) / GAP - 1);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int i = NM / GAP; i < NP; i++) stage_piece(t, nxt, i);  // more DMA instructions than MFMA gaps (thin tiles)
        if constexpr (LW == 0) stage_advance();
      } else {
        if constexpr (LW == 0) { if (kt + NS - 1 < nk) stage(kt + NS - 1, nxt); }
        compute(cur);
      }
      cur = (cur + 1 == NS) ? 0 : cur + 1;
      nxt = (nxt + 1 == NS) ? 0 : nxt + 1;
    }
  } else {
    static_assert(!PP || (NW == 8 && NS == 3), "ping-pong schedule: 8 waves, 3-slot ring");
    const bool grpB = wave >= 4;
    // wait until at most `n` of this wave's most recent K-tiles are still in flight
    auto wait_in_flight = [&](int n) {
      wait_alt_begin();
      if (n >= 2) { if (lps_hi) wait_vmcnt<2 * LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<2 * LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
      else if (n == 1) { if (lps_hi) wait_vmcnt<LPS_HI, LPS_HI, 0, LPS_ALT>(); else wait_vmcnt<LPS_LO, LPS_LO, 0, LPS_ALT2>(); }
      else wait_vmcnt<0>();
      wait_alt_end();
    };
    auto phase_barrier = [&]() {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    // prologue: tiles 0..2 in flight
    const int npro = min(3, nk);
    for (int s = 0; s < npro; s++) stage(s, s);
    wait_in_flight(npro - 1);  // own share of tile 0 landed
    phase_barrier();           // tile 0 complete in LDS
    // Two specialised loops (same barrier count) so each group's fragment registers have one live range.
    if (!grpB) {
      read_frags(0);                       // "phase Y(-1)"
      phase_barrier();
      int slot = 0;
      for (int t = 0; t < nk; t++) {
        const int slot1 = slot == 2 ? 0 : slot + 1;
        mfma_all();                                                   // X(t): multiply tile t (B reads tile t)
        if (t + 1 < nk) wait_in_flight(t + 2 < nk ? 1 : 0);            // own share of tile t+1 landed
        phase_barrier();                                              // B1(t): slot of tile t free, tile t+1 complete
        if (t + 3 < nk) stage(t + 3, slot);                           // Y(t): refill, then pick up tile t+1
        if (t + 1 < nk) read_frags(slot1);
        phase_barrier();                                              // B2(t)
        slot = slot1;
      }
    } else {
      phase_barrier();
      int slot = 0;
      for (int t = 0; t < nk; t++) {
        read_frags(slot);                                             // X(t): read tile t (A multiplies tile t)
        if (t + 1 < nk) wait_in_flight(t + 2 < nk ? 1 : 0);
        phase_barrier();                                              // B1(t)
        if (t + 3 < nk) stage(t + 3, slot);                           // Y(t): refill, multiply tile t
        mfma_all();
        phase_barrier();                                              // B2(t)
        slot = slot == 2 ? 0 : slot + 1;
      }
    }
  }

  TS_MARK(2);
  // ---- split-K hand-off ---------------------------------------------------------------------
  // Placement-independent publish (cdna_hip_programming.md, split-K rule 2): the partial tile leaves through sc1
  // (write-through to memory) buffer stores, every storing wave drains its stores, one relaxed agent-scope flag store
  // publishes it; the consumer polls relaxed and reads the partial back with sc1 loads.  No device-scope fence
  // (`__threadfence()` = L2 write-back per producer block made split-K a net loss: 157 vs 160.5 steps/s).  Keeping the
  // pair on one XCD is only a speed choice.
  if (S > 1) {
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    // partial tile in the accumulator layout: (fragment, lane) -> 16 B, so every wave instruction moves 1 KiB contiguous
    const int tile = tm * p.tiles_n + tn;
    if (ks) {  // producer: slot ks-1 of this tile
      float* wsw = p.sk_ws + (((long long)(ks - 1) * ntile + tile) * NW + wave) * (FM * FN * 256);
      const rsrc_t rws = make_rsrc(wsw, FM * FN * 1024);
      wait_vmcnt<0>();  // dead tail DMA has landed before the wave may end
      tsd_jitter();
#pragma unroll
      for (int a = 0; a < FM; a++)
#pragma unroll
        for (int b = 0; b < FN; b++)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc[a][b]), rws, ((a * FN + b) * 64 + lane) * 16, 0, /*sc1*/ 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains before the flag
      __syncthreads();
      if (tid == 0) __hip_atomic_store(p.sk_flags + (ks - 1) * ntile + tile, p.sk_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
    // Flags carry the launch's epoch (a per-context counter, never 0) and are never reset: a producer of an EARLIER launch
    // that lands late cannot arm this launch's hand-off, and a consumer that gives up leaves nothing armed behind it.
    for (int sl = 1; sl < S; sl++) {  // consumer: fixed order slice 0 + slice 1 + ... (bitwise reproducible)
      int* flag = p.sk_flags + (sl - 1) * ntile + tile;
      if (tid == 0) {  // bounded relaxed poll (producers have the lower block ids: always resident first)
        int spins = 0;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != p.sk_epoch && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(4);
        // timed out: counted in a sticky per-context word that tsd_ctx_synchronize / the session downloads turn into TSD_E_STATE
        if (spins >= (1 << 20)) atomicAdd(&p.sk_flags[4095], 1);
      }
      __syncthreads();
      float* wsw = p.sk_ws + (((long long)(sl - 1) * ntile + tile) * NW + wave) * (FM * FN * 256);
      const rsrc_t rws = make_rsrc(wsw, FM * FN * 1024);
#pragma unroll
      for (int a = 0; a < FM; a++)
#pragma unroll
        for (int b = 0; b < FN; b++) {
          const u4 v = __builtin_amdgcn_raw_buffer_load_b128(rws, ((a * FN + b) * 64 + lane) * 16, 0, /*sc1*/ 16);
          acc[a][b] += __builtin_bit_cast(f4, v);
        }
    }
  }
  // ---- epilogue ---------------------------------------------------------------------------
  tsd_jitter();
  const int g = lane >> 4;
  // epilogue terms that only one kernel family uses are compiled out of the other (launch_gemm rejects the combinations):
  // GEGLU / per-row bias exist for dense GEMMs only, the time-embedding row vector / upsampled residual for conv3x3 only
  const int epi = p.epi & (CONV ? ~(EPI_GEGLU | EPI_BIAS_M) : ~(EPI_ROWVEC | EPI_RES_UPS));
  const int nb = n0 + wn * BNw + g * (4 * FN);

  // Coalesced path (every tile shape but the thin N <= 16 one, N % 8 == 0).  The MFMA layout leaves each lane with
  // 4*FN consecutive columns of one row, i.e. 8-byte pieces 8*FN bytes apart across the wave - store-issue bound
  // when written directly.  Instead each wave transposes 32 rows at a time through its own slice of the (now idle)
  // LDS ring in fp32, after the per-column terms, and then walks them row-major: every lane owns 8 consecutive
  // columns, the residual arrives as one 16-B load, the result leaves as one 16-B store, and a wave instruction
  // covers whole BNw*2-byte row segments.
  if constexpr (FN >= 4 && FM % 2 == 0) {
    if ((p.N & 7) == 0) {
      constexpr int EPP = BNw + 4;               // LDS row pitch in floats (16-B aligned, breaks the power of two)
      constexpr int CH = BNw / 8;                // 8-column chunks per row
      constexpr int ITEMS = 32 * CH / 64;
      static_assert((32 * CH) % 64 == 0, "chunks must divide evenly among the lanes");
      static_assert(NW * 32 * EPP * 4 <= NS * TILE_BYTES, "epilogue staging must fit in the tile ring");
      // The residual tile is fetched NOW, in the row-major item layout, so its HBM/L2 round trip overlaps the DMA
      // drain, the barrier and the LDS transposes instead of stalling every pass (measured with -DTSD_GEMM_TS: the
      // residual cost 6 us of a 21 us 32768x320x320 GEMM when loaded inside the item loop).  Loads are unconditional
      // (clamped addresses) so exactly RES_LOADS VMEM instructions are younger than the last DMA.
      constexpr int RES_LOADS = (FM / 2) * ITEMS;
      h8 resv[FM / 2][ITEMS];
      // the lane's column biases are fetched here too (older than the residual loads, so the counted wait below covers them): loaded
      // inside the pass loop their L2 / HBM round trip sat in front of the first transpose of every block
      // (not in the 12-wave staggered tiles: 168 registers per wave, the 20 extra ones spill)
      constexpr bool PREBIAS = !(LW > 0 && NW == 8);
      f4 bvn[FN];
#pragma unroll
      for (int b = 0; b < FN; b++) bvn[b] = f4{0.f, 0.f, 0.f, 0.f};
      if (PREBIAS && (epi & EPI_BIAS_N)) {
#pragma unroll
        for (int b = 0; b < FN; b++) bvn[b] = *(const f4*)(p.bias + min(nb + b * 4, p.N - 4));
      }
      if (epi & EPI_RESIDUAL) {
        const half_t* rbase = p.R + (long long)bz * p.sR;
#pragma unroll
        for (int pass = 0; pass < FM / 2; pass++)
#pragma unroll
          for (int i = 0; i < ITEMS; i++) {
            const int t = lane + 64 * i;
            const int r = t / CH, c = t - r * CH;
            const int m = min(m0 + wm * BMw + pass * 32 + r, p.M - 1);
            const int n = min(n0 + wn * BNw + c * 8, p.N - 8);
            long long rrow = m;
            if (epi & EPI_RES_UPS) {
              const int hw = p.Ho * p.Wo;
              const int bb = m / hw, rem = m - bb * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
              rrow = ((long long)bb * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
            }
            resv[pass][i] = *(const h8*)(rbase + rrow * p.ldr + n);
          }
        wait_vmcnt<RES_LOADS, 0, RES_LOADS>();      // every DMA (including the dead tail tiles) has landed ...
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();    // ... and every wave is done reading the ring
      asm volatile("" ::: "memory");
      TS_MARK(3);
      float* ep = (float*)smem + wave * (32 * EPP);
      // EPI_GNSTATS: per-channel (sum, sumsq) of the ROUNDED output over this wave's rows, lane j < BNw/2 owning columns
      // 2j and 2j+1; reduced to the consumer GroupNorm's groups at the end (replaces its statistics pass).
      const bool gn_on = (epi & EPI_GNSTATS) && m0 + wm * BMw < p.M;
#pragma unroll
      for (int pass = 0; pass < FM / 2; pass++) {
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          const int a = pass * 2 + hf;
          const int m = m0 + wm * BMw + a * 16 + rsel;
          float v[4 * FN];
#pragma unroll
          for (int b = 0; b < FN; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) v[b * 4 + r] = acc[a][b][r] * p.out_scale;
          if (epi & EPI_BIAS_M) {
            const float bm = p.bias[m < p.M ? m : p.M - 1];
#pragma unroll
            for (int j = 0; j < 4 * FN; j++) v[j] += bm;
          }
          // Column terms are loaded branch-free (clamped address, columns beyond N are never stored): a per-lane branch around each
          // f4 made the compiler carry whole copies of v[] across the divergent regions - 3848 scratch accesses in this kernel
          if (epi & EPI_BIAS_N) {
#pragma unroll
            for (int b = 0; b < FN; b++) {
              const f4 bv = PREBIAS ? bvn[b] : *(const f4*)(p.bias + min(nb + b * 4, p.N - 4));
#pragma unroll
              for (int r = 0; r < 4; r++) v[b * 4 + r] += bv[r];
            }
          }
          if (epi & EPI_ROWVEC) {
            const float* rv = p.rowvec + (long long)((m < p.M ? m : p.M - 1) / p.rows_per_batch) * p.rowvec_ld;
#pragma unroll
            for (int b = 0; b < FN; b++) {
              const f4 bv = *(const f4*)(rv + min(nb + b * 4, p.N - 4));
#pragma unroll
              for (int r = 0; r < 4; r++) v[b * 4 + r] += bv[r];
            }
          }
          float* row = ep + (hf * 16 + rsel) * EPP + g * (4 * FN);
#pragma unroll
          for (int b = 0; b < FN; b++) *(f4*)(row + b * 4) = f4{v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]};
        }
        // same-wave LDS operations complete in order: the row-major reads below see the stores above
        if (!CONV && p.vt && n0 >= p.vt_n0) {
          // transposed tail (block-uniform: vt_n0 is a multiple of the tile width): each lane takes one column and eight consecutive
          // rows of the pass - 16 B of V^T[b][channel][token]; four lanes cover the pass's 32 tokens of a channel (64 B)
#pragma unroll
          for (int i = 0; i < (BNw * 4 + 63) / 64; i++) {
            const int t = lane + 64 * i;
            const int col = t >> 2, rc = t & 3;
            const int m = m0 + wm * BMw + pass * 32 + rc * 8;
            const int n = n0 + wn * BNw + col;
            if (t >= BNw * 4 || m >= p.M || n >= p.N) continue;
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (half_t)ep[(rc * 8 + j) * EPP + col];
            const int bsm = m / p.vt_S, srow = m - bsm * p.vt_S;
            *(h8*)(p.vt + (long long)bsm * p.vt_sB + (long long)(n - p.vt_n0) * p.vt_ld + srow) = o;
          }
          continue;
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
          const int t = lane + 64 * i;
          const int r = t / CH, c = t - r * CH;
          const int m = m0 + wm * BMw + pass * 32 + r;
          const int n = n0 + wn * BNw + c * 8;
          const f4 x0 = *(const f4*)(ep + r * EPP + c * 8), x1 = *(const f4*)(ep + r * EPP + c * 8 + 4);
          if (m >= p.M || n >= p.N) continue;
          float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
          if (epi & EPI_RESIDUAL) {
            const h8 rv = resv[pass][i];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] += (float)rv[j];
          }
          if (epi & EPI_GEGLU) {
            h4 o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (half_t)(v[2 * j] * gelu_tanh_f(v[2 * j + 1]));
            *(h4*)((half_t*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + (n >> 1)) = o;
          } else if (epi & EPI_OUT_F32) {
            float* cp = (float*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + n;
            *(f4*)cp = f4{v[0], v[1], v[2], v[3]};
            *(f4*)(cp + 4) = f4{v[4], v[5], v[6], v[7]};
          } else {
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = (half_t)v[j];
            *(h8*)((half_t*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + n) = o;
            if (gn_on) *(h8*)(ep + r * EPP + c * 8) = o;  // back into the slot just read (same-wave LDS ops are ordered)
          }
        }
        if (gn_on) {
          // one 32-row slab per pass, whatever the tile shape: the partials (and so the normalised tensor) are bitwise
          // independent of the tile configuration, hence of the batch size
          float cs1[2] = {0.f, 0.f}, cs2[2] = {0.f, 0.f};
          if (lane < BNw / 2) {
            const char* colp = (const char*)(ep + (lane >> 2) * 8) + (lane & 3) * 4;
#pragma unroll 8
            for (int r = 0; r < 32; r++) {
              const h2 hv = *(const h2*)(colp + r * (EPP * 4));
              const float f0 = (float)hv[0], f1 = (float)hv[1];
              cs1[0] += f0; cs2[0] += f0 * f0;
              cs1[1] += f1; cs2[1] += f1 * f1;
            }
          }
          float* cst = ep;  // [BNw][2] (the transposed rows are dead now)
          if (lane < BNw / 2) *(f4*)(cst + 4 * lane) = f4{cs1[0], cs2[0], cs1[1], cs2[1]};
          const int mrow = m0 + wm * BMw + pass * 32;
          const int b = mrow / p.gn_hw, slab = (mrow - b * p.gn_hw) >> 5;
          float* obase = p.gn_part + ((long long)b * p.gn_nslab + slab) * p.gn_G * 2;
          for (int gi = lane; gi * p.gn_cpg < BNw; gi += 64) {  // up to BNw groups in the slice (one channel per group)
            const int colbase = n0 + wn * BNw + gi * p.gn_cpg;
            if (colbase >= p.N) break;
            float s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < p.gn_cpg; k++) {
              s1 += cst[(gi * p.gn_cpg + k) * 2];
              s2 += cst[(gi * p.gn_cpg + k) * 2 + 1];
            }
            *(f2*)(obase + (colbase / p.gn_cpg) * 2) = f2{s1, s2};
          }
        }
      }
#ifdef TSD_GEMM_TS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TS_MARK(4);
#endif
      return;
    }
  }
#pragma unroll
  for (int a = 0; a < FM; a++) {
    __builtin_amdgcn_sched_barrier(0);  // one fragment row at a time (this rarely taken path must not set the kernel's scratch size)
    const int m = m0 + wm * BMw + a * 16 + rsel;
    if (m >= p.M) continue;
    float v[4 * FN];
#pragma unroll
    for (int b = 0; b < FN; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) v[b * 4 + r] = acc[a][b][r] * p.out_scale;
    if (epi & EPI_BIAS_M) {
      const float bm = p.bias[m];
#pragma unroll
      for (int j = 0; j < 4 * FN; j++) v[j] += bm;
    }
    // column terms branch-free (clamped address; columns beyond N are never stored) - see the coalesced path
    if (epi & EPI_BIAS_N) {
#pragma unroll
      for (int b = 0; b < FN; b++) {
        const f4 bv = *(const f4*)(p.bias + min(nb + b * 4, p.N - 4));
#pragma unroll
        for (int r = 0; r < 4; r++) v[b * 4 + r] += bv[r];
      }
    }
    if (epi & EPI_ROWVEC) {
      const float* rv = p.rowvec + (long long)(m / p.rows_per_batch) * p.rowvec_ld;
#pragma unroll
      for (int b = 0; b < FN; b++) {
        const f4 bv = *(const f4*)(rv + min(nb + b * 4, p.N - 4));
#pragma unroll
        for (int r = 0; r < 4; r++) v[b * 4 + r] += bv[r];
      }
    }
    if (epi & EPI_RESIDUAL) {
      long long rrow = m;
      if (epi & EPI_RES_UPS) {
        const int hw = p.Ho * p.Wo;
        const int bb = m / hw, rem = m - bb * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
        rrow = ((long long)bb * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
      }
      const half_t* rp = p.R + (long long)bz * p.sR + rrow * p.ldr;
#pragma unroll
      for (int b = 0; b < FN; b++) {
        const h4 rvv = *(const h4*)(rp + min(nb + b * 4, p.N - 4));
#pragma unroll
        for (int r = 0; r < 4; r++) v[b * 4 + r] += (float)rvv[r];
      }
    }
    if (epi & EPI_GEGLU) {
      half_t* cp = (half_t*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + (nb >> 1);
#pragma unroll
      for (int b = 0; b < FN; b++)
        if (nb + b * 4 < p.N) {
          h2 o;
          o[0] = (half_t)(v[b * 4 + 0] * gelu_tanh_f(v[b * 4 + 1]));
          o[1] = (half_t)(v[b * 4 + 2] * gelu_tanh_f(v[b * 4 + 3]));
          *(h2*)(cp + b * 2) = o;
        }
    } else if (epi & EPI_OUT_F32) {
      float* cp = (float*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + nb;
#pragma unroll
      for (int b = 0; b < FN; b++)
        if (nb + b * 4 < p.N) *(f4*)(cp + b * 4) = f4{v[b * 4], v[b * 4 + 1], v[b * 4 + 2], v[b * 4 + 3]};
    } else {
      half_t* cp = (half_t*)p.C + (long long)bz * p.sC + (long long)m * p.ldc + nb;
#pragma unroll
      for (int b = 0; b < FN; b++)
        if (nb + b * 4 < p.N) {
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (half_t)v[b * 4 + r];
          *(h4*)(cp + b * 4) = o;
        }
    }
  }
}

// ---- host side ------------------------------------------------------------------------------
template <int WGM, int WGN, int FM, int FN, bool CONV, int NS = 2, bool PP = false, int CV = 0, int LW = 0>
static int launch_cfg(tsd_ctx* ctx, const GemmK& k, int batch) {
  constexpr int BM = WGM * FM * 16, BN = WGN * FN * 16;
  constexpr bool HX = CV == 1;
  constexpr int LDS = HX ? NS * BN * 128 + 2 * (BM / 8 + 1) * 1024 : NS * (BM + BN) * 128;
  auto fn = gemm_kernel<WGM, WGN, FM, FN, CONV, NS, PP, CV, LW>;
  static std::atomic<unsigned long long> attr_set{0};  // per DEVICE: the attribute is stored per device (one bit each; setting it twice is harmless)
  if (!((attr_set.load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
    if (ctx->opt.debug_occ) {
      int nb = -1;
      hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, (WGM * WGN + LW) * 64, LDS);
      fprintf(stderr, "[occ] gemm<%d,%d,%d,%d,%d,%d,lw%d> LDS=%d blocks/CU=%d (%s)\n", WGM, WGN, FM, FN, (int)CONV, NS, LW, LDS, nb,
              hipGetErrorString(e));
    }
  }
  GemmK kk = k;
#ifdef TSD_GEMM_TS
  kk.ts = g_ts;
#endif
  kk.tiles_n = ceil_div(k.N, BN);
  const int tiles_m = ceil_div(k.M, BM);
  {  // XCD grid: minimise W_bytes * xm + A_bytes * xn over the factorizations of 8 that divide the tile grid
    const int force = ctx->opt.xcdn;
    const double wb = 2.0 * k.N * k.K;
    const double ab = CONV ? 2.0 * (k.M / (k.Ho * k.Wo)) * k.Hs * k.Ws * k.Cin : 2.0 * (double)k.M * k.K;
    int best = 1;
    double best_cost = wb * 8 + ab;
    for (int xn = 2; xn <= 8; xn *= 2) {
      const int xm = 8 / xn;
      if (kk.tiles_n % xn || tiles_m % xm) continue;
      const double cost = wb * xm + ab * xn;
      if (cost < best_cost) { best_cost = cost; best = xn; }
    }
    if (force > 0) best = (kk.tiles_n % force == 0 && tiles_m % (8 / force) == 0) ? force : 1;
    kk.xcd_n = best;
  }
  dim3 grid(tiles_m * kk.tiles_n * (kk.splitk > 1 ? kk.splitk : 1), batch);
  hipLaunchKernelGGL(fn, grid, dim3((WGM * WGN + LW) * 64), LDS, ctx->stream, kk);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// tile configurations: {wave grid M x N, fragments per wave M x N, LDS ring depth}
//  id  tile      stages  LDS      use
//   0  128x160   2       72 KiB   big-M UNet widths (N % 160 == 0), 2 blocks/CU
//   1   64x160   2       56 KiB   mid-M
//   2  128x128   2       64 KiB   VAE widths
//   3   64x128   2       48 KiB
//   4  128x16    2       36 KiB   N <= 16, many tiles (decoder 128->3)
//  24   64x16    4       40 KiB   N <= 16, few tiles (UNet 320->4)
//   5  128x160   3      108 KiB   deep ring, 1 block/CU
//   6   64x160   4      112 KiB   few-tile problems (M = 2048 level)
//   7   64x160   3       84 KiB
//   8  128x128   3       96 KiB
//   9   64x128   4       96 KiB
//  10   64x128   3       72 KiB
//  11  256x160   3      156 KiB   8 waves, 1 block/CU, two K-tiles of DMA in flight
//  12  256x160   2      104 KiB   8 waves
//  13  256x128   3      144 KiB   8 waves
constexpr int N_GEMM_CFG = 56;

// SKV = 2 for a conv3x3 with a fused 1x1 skip source (GemmK::Cin1 > 0), else 0: the plain kernels carry none of its scalar state
template <bool CONV, int SKV = 0>
static int launch_by_id(tsd_ctx* ctx, const GemmK& k, int batch, int id) {
  if constexpr (CONV && SKV == 0) {
    if (k.Cin1 > 0) return launch_by_id<CONV, 2>(ctx, k, batch, id);
  }
  switch (id) {
    case 0: return launch_cfg<2, 2, 4, 5, CONV, 2, false, SKV>(ctx, k, batch);
    case 1: return launch_cfg<2, 2, 2, 5, CONV, 2, false, SKV>(ctx, k, batch);
    case 2: return launch_cfg<2, 2, 4, 4, CONV, 2, false, SKV>(ctx, k, batch);
    case 3: return launch_cfg<2, 2, 2, 4, CONV, 2, false, SKV>(ctx, k, batch);
    case 4: return launch_cfg<4, 1, 2, 1, CONV, 2, false, SKV>(ctx, k, batch);
    case 5: return launch_cfg<2, 2, 4, 5, CONV, 3, false, SKV>(ctx, k, batch);
    case 6: return launch_cfg<2, 2, 2, 5, CONV, 4, false, SKV>(ctx, k, batch);
    case 7: return launch_cfg<2, 2, 2, 5, CONV, 3, false, SKV>(ctx, k, batch);
    case 8: return launch_cfg<2, 2, 4, 4, CONV, 3, false, SKV>(ctx, k, batch);
    case 9: return launch_cfg<2, 2, 2, 4, CONV, 4, false, SKV>(ctx, k, batch);
    case 10: return launch_cfg<2, 2, 2, 4, CONV, 3, false, SKV>(ctx, k, batch);
    case 11: return launch_cfg<4, 2, 4, 5, CONV, 3, false, SKV>(ctx, k, batch);
    case 13: return launch_cfg<4, 2, 4, 4, CONV, 3, false, SKV>(ctx, k, batch);
    // thin tile (N <= 16) with 64 rows and a 4-slot ring: few-tile problems (the UNet's 320 -> 4 output convolution)
    case 24: return launch_cfg<4, 1, 1, 1, CONV, 4, false, SKV>(ctx, k, batch);
    // loader-wave variants (LW = 4): 40 + the id of the 4-wave one-block-per-CU configuration they extend, 51 = cfg 11 + loaders
    case 45: return launch_cfg<2, 2, 4, 5, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 46: return launch_cfg<2, 2, 2, 5, CONV, 4, false, SKV, 4>(ctx, k, batch);
    case 47: return launch_cfg<2, 2, 2, 5, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 48: return launch_cfg<2, 2, 4, 4, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 49: return launch_cfg<2, 2, 2, 4, CONV, 4, false, SKV, 4>(ctx, k, batch);
    case 50: return launch_cfg<2, 2, 2, 4, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 51: return launch_cfg<4, 2, 4, 5, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 53: return launch_cfg<4, 2, 4, 4, CONV, 3, false, SKV, 4>(ctx, k, batch);
    case 54: return launch_cfg<4, 2, 2, 5, CONV, 3, false, SKV, 4>(ctx, k, batch);  // 128x160, staggered: 8 compute waves of 32x80
    case 55: return launch_cfg<4, 2, 2, 4, CONV, 3, false, SKV, 4>(ctx, k, batch);  // 128x128
    // halo-x variants of 0 and 2 (conv3x3, stride 1: hx_eligible)
    case 30: if constexpr (CONV && SKV == 0) return launch_cfg<2, 2, 4, 5, CONV, 2, false, 1>(ctx, k, batch); else break;
    case 32: if constexpr (CONV && SKV == 0) return launch_cfg<2, 2, 4, 4, CONV, 2, false, 1>(ctx, k, batch); else break;
#ifdef TSD_GEMM_EXPERIMENTAL  // measured, not faster: built only to reproduce those numbers
    case 12: return launch_cfg<4, 2, 4, 5, CONV, 2, false, SKV>(ctx, k, batch);
    case 14: return launch_cfg<2, 2, 8, 5, CONV, 3, false, SKV>(ctx, k, batch);
    case 15: return launch_cfg<2, 2, 8, 5, CONV, 2, false, SKV>(ctx, k, batch);
    case 16: return launch_cfg<4, 2, 4, 5, CONV, 3, true, SKV>(ctx, k, batch);  // 256x160 ping-pong
    case 17: return launch_cfg<4, 2, 2, 5, CONV, 3, true, SKV>(ctx, k, batch);  // 128x160 ping-pong
    case 18: return launch_cfg<4, 2, 4, 4, CONV, 3, true, SKV>(ctx, k, batch);  // 256x128 ping-pong
    case 19: return launch_cfg<4, 2, 2, 4, CONV, 3, true, SKV>(ctx, k, batch);  // 128x128 ping-pong
    case 20: return launch_cfg<2, 2, 2, 5, CONV, 5, false, SKV>(ctx, k, batch);  // 64x160, 5-slot ring: slower than 4 slots (667 vs 821 TF)
    case 21: return launch_cfg<2, 2, 2, 4, CONV, 6, false, SKV>(ctx, k, batch);  // 64x128, 6-slot ring
#endif
    default: break;
  }
  TSD_FAIL(TSD_E_ARG, "gemm: unknown tile configuration %d", id);
}

// conv3x3 problems the halo-x K order can run: stride 1 on the source grid, whole 128-pixel tiles made of 64- or 128-pixel
// image-row segments, no split-K
// OFF by default: the halo-x order sums K in a different order than every other tile configuration, and which
// configuration runs depends on M - a sample computed alone would no longer equal its row of a batch bit for bit.  Measured
// with it on (TSD_CONV_HALO=1: the 128x128-tile convs, 2: the 128x160 ones too): decoder 26.1 -> 25.7 ms, encoder 13.95 ->
// 13.70 ms, UNet step unchanged.  tests/test_gpu_ops.py keeps the path correct against the plain configurations.
static bool hx_shape_ok(const GemmK& k);
static bool hx_eligible(const TsdOptions& o, const GemmK& k) { return o.conv_halo > 0 && hx_shape_ok(k); }
static bool hx_shape_ok(const GemmK& k) {
  return k.stride == 1 && !k.ups && k.pad == 1 && k.splitk <= 1 && !k.Cin1 && k.Hs == k.Ho && k.Ws == k.Wo && k.Cin % 64 == 0 &&
         (k.Wo == 64 || k.Wo % 128 == 0) && ((long long)k.Ho * k.Wo) % 128 == 0 && k.M % 128 == 0 && k.Ho < 2040 && k.Wo < 2040;
}

// Split-K by 2 pays when a 128-row tiling leaves about half the CUs idle and K is long: the M = 2048 level of the UNet
// (16 x 8 tiles of 128x160, K = 5120..23040).  64-row tiles fill the chip there but move 46 flop per LDS-DMA byte and
// are bound by the per-CU DMA rate; two 128-row half-K blocks move 71 flop/B.
// Workgroup -> XCD placement probe (HW_REG_XCC_ID).  The split-K hand-off is placement-independent (sc1 stores/loads),
// so this only tells whether the same-XCD pairing of its two blocks - a speed choice - holds on this device.
__global__ void k_probe_xcc(int* out) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)(x & 0xf);
}
static bool xcd_round_robin() {
  static std::atomic<int> cached{-1};  // a probe result: every thread computes the same value
  if (cached.load(std::memory_order_relaxed) >= 0) return cached.load(std::memory_order_relaxed) != 0;
  int ok = 0;
  const int nb = 1024;
  int* d = nullptr;
  if (hipMalloc((void**)&d, nb * sizeof(int)) != hipSuccess) return false;
  std::vector<int> h(nb, -1);
  hipLaunchKernelGGL(k_probe_xcc, dim3(nb), dim3(64), 0, 0, d);
  if (hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost) == hipSuccess) {
    ok = 1;
    for (int b = 0; b < nb; b++) if (h[b] < 0 || h[b] != h[b & 7]) ok = 0;
    for (int i = 0; i < 8; i++) for (int j = 0; j < i; j++) if (h[i] == h[j]) ok = 0;
  }
  hipFree(d);
  cached.store(ok, std::memory_order_relaxed);
  return ok != 0;
}
extern "C" int tsd_debug_xcd_round_robin(void) { return xcd_round_robin() ? 1 : 0; }

// slices for K >= 8192 at the 16x16 level asked for by the graph being enqueued on this context (0: default)
int gemm_set_splitk_big(tsd_ctx* ctx, int ways) { const int prev = ctx->opt.sk_big_graph; ctx->opt.sk_big_graph = ways; return prev; }
// Split-K plan: number of K slices (1 = none) and the tile configuration the split launch runs with.
static int splitk_plan(const TsdOptions& o, int M, int N, int K, int batch, int rps, int* cfg) {
  const int on = o.splitk;
  // The decision must not depend on the batch size (bitwise batch invariance: a split changes the fp32 summation
  // tree), so it keys on the layer: rows per sample, N and K.
  //  * rps <= 256 (the 16x16 level of a 64x64 latent): 2 slices of 128-row tiles once K >= 4096;
  //  * rps <= 64 (an 8x8 level: the full-size UNet's deepest at a 64x64 latent, the 23-layer graph's at 32x32): M is
  //    a few hundred rows, so 64-row tiles and up to 8 slices - 32 tiles x 8 fill the chip where 16 tiles x 2 left
  //    7/8 of it idle.
  const int min_k = o.splitk_mink, max_tiles = o.splitk_tiles, small_ways = o.splitk_small;
  // Round 3 (late): two more layer classes that left half the chip idle at batch 8 -
  //  * rps <= 256 with at most 4 tile columns (the 32x32 -> 16x16 downsampling conv, N = 640, K = 5760: 64 tiles): 4 slices;
  //  * rps <= 1024 with N * rps <= 320 * 1024 (the 64x64 -> 32x32 downsampling conv, N = 320, K = 2880: 128 tiles): 2 slices.
  const int wide = o.splitk_wide;
  const bool mid = wide && rps > 256 && rps <= 1024 && (long long)N * rps <= 320LL * 1024 && K >= 2880;
  // Round 4 (TSD_GEMM_SK256): 256x160 tiles (the staggered loader-wave configuration 51) for split launches, with twice the slices so
  // that the grid stays the same: a K tile then pulls 52 KB from the L2 for twice the products of a 128x160 tile's 36 KB (98 instead of
  // 71 flop per L2 byte; the K loops of these layers run on the L2 -> CU path).  Bit 0: the 16x16-level layers that split
  // already; bit 1: the 32x32-level 640-wide convolutions (K >= 5760: 256 tiles of 128x160 today, no split).  Keyed on the layer only.
  const bool mid256 = (o.sk256 & 2) && rps == 1024 && N == 640 && K >= 5760;
  if (!on || batch != 1 || N <= 16 || rps <= 0 || (rps > 256 && !mid && !mid256)) return 1;
  const bool n160 = (N % 160 == 0);
  const int BN = n160 ? 160 : 128;
  int ways = 1, BM = 128;
  if (rps <= 64 && small_ways > 1) {
    ways = K >= 8192 ? 8 : (K >= 2048 ? 4 : (K >= 1024 ? 2 : 1));
    if (ways > small_ways) ways = small_ways;
    BM = 64;
    const int deep = o.splitk_ring4;
    if (cfg) *cfg = deep ? (n160 ? 6 : 9) : (n160 ? 7 : 10);
  } else {
    const int big_env = o.splitk_big;
    // 2 slices for the 23-layer UNet at batch 8; the full-size UNet's graph asks for 4 (gemm_set_splitk_big: +4.3 % at its batch of 4,
    // -0.8 % on the headline).  A per-GRAPH choice, so every batch size of a model sums in the same tree.
    const int big_ways = big_env ? big_env : (o.sk_big_graph ? o.sk_big_graph : 2);
    ways = K >= 8192 ? big_ways : (K >= min_k ? 2 : 1);
    if (wide && ways == 2 && ceil_div(N, BN) <= 4) ways = 4;
    if (mid) ways = 2;
    const int sk128 = o.sk_cfg;  // 45: the same tile with loader waves
    if (cfg) *cfg = n160 ? sk128 : 8;
    const int tn = ceil_div(N, BN);
    const bool xcd256 = tn % 8 == 0 || (wide && rps % 256 == 0 && (tn * (rps / 256)) % 8 == 0);  // a tile's slices stay on one XCD
    if (n160 && xcd256 && ((mid256 && !mid) || ((o.sk256 & 1) && rps == 256 && ways >= 2 && ways <= 4))) {
      ways = mid256 ? 2 : ways * 2;
      BM = 256;
      if (cfg) *cfg = 51;
    }
  }
  // Eligibility looks at N only (8 | N-tiles keeps a tile's slices on one XCD for any M): a condition on the tile
  // count would make the split - and with it the fp32 summation tree - depend on the batch.  The two M-dependent
  // guards below cannot trigger inside the API's limits (B <= 16 with rps <= 256 gives at most 256 tiles).
  // (round 3: 8 | tiles of one sample does the same for the layers with fewer tile columns)
  const int tiles_n = ceil_div(N, BN), tiles = ceil_div(M, BM) * tiles_n;
  const bool xcd_ok = tiles_n % 8 == 0 || (wide && rps % BM == 0 && (tiles_n * (rps / BM)) % 8 == 0);
  if (ways == 1 || !xcd_ok || tiles > 2 * max_tiles || (ways - 1) * tiles > 4095) return 1;
  return ways;
}
static int choose_cfg(const TsdOptions& o, int M, int N, int K, int batch, bool conv, int rps = 0) {
  {  // tuning aid: TSD_GEMM_CFG_OVERRIDE="M,N,K:cfg[;M,N,K:cfg...]" forces a tile configuration for exact shapes inside a real step
    const char* ov = o.cfg_override;
    if (ov[0]) {
      for (const char* q = ov; q && *q;) {
        int m = 0, n = 0, k = 0, c = 0;
        if (sscanf(q, "%d,%d,%d:%d", &m, &n, &k, &c) == 4 && m == M && n == N && k == K) return c;
        q = strchr(q, ';');
        if (q) q++;
      }
    }
  }
  if (N <= 16) {
    // the UNet's 320 -> 4 output convolution is one 128-row block per CU walking 45 K tiles behind a 2-slot ring: 64-row tiles with
    // a 4-slot ring (two blocks per CU, three tiles in flight) take 20 us where it took 33 in the step; with thousands of tiles
    // (the decoder's 128 -> 3 at 512 x 512) the 128-row tile stays ahead (277 vs 329 us).  Bitwise the same results either way.
    const int thin = o.thin_cfg;
    if (thin) return thin;
    return (long long)ceil_div(M, 128) * batch <= 1024 ? 24 : 4;
  }
  {
    int sk_cfg = 0;
    if (splitk_plan(o, M, N, K, batch, rps, &sk_cfg) > 1) return sk_cfg;
  }
  const bool n160 = (N % 160 == 0);
  const int BN = n160 ? 160 : 128;
  // measured on MI355X (scripts/bench_gemm.py with REAL_EPI=1, pinned issue order):
  //  * >= 2 tiles of 128 rows per CU: two 4-wave blocks per CU (cfg 0/2);
  //  * dense GEMMs whose 256x160 tiling is exactly one or two full rounds of the 256 CUs: the 8-wave tile (cfg 11)
  //    halves the operand traffic per flop and its prologue/epilogue count;
  //  * around one 128-row tile per CU: long K -> one 128-row block with a 3-slot DMA ring (one wave per SIMD, the
  //    pinned schedule keeps its MFMA pipe fed), short K -> 64-row tiles, two blocks per CU (fixed costs overlap);
  //  * fewer tiles than that: 64-row tiles with 3 ring slots, 4 when K is long.
  const long long t128 = (long long)ceil_div(M, 128) * ceil_div(N, BN) * batch;
  const long long t256 = (long long)ceil_div(M, 256) * ceil_div(N, BN) * batch;
  const int tune = o.tune;  // A/B switch for the rules below
  // Round 3: the staggered wave-specialised 256-row tiles (51 / 53: 8 compute waves in two groups + 4 loader waves) where a
  // 256-row tiling gives every CU whole tiles - measured -5...-10 % against configurations 0 / 2 / 11 on these shapes
  // (profiles/r03_loader_waves_ab.txt); TSD_GEMM_TUNE bit 2 turns them off.  Results are bitwise those of every other tile.
  if ((tune & 4) && M % 256 == 0) {
    if (n160 && conv && t256 >= 256 && t256 % 256 == 0) return 51;
    if (n160 && !conv && K >= 256 && (t256 == 256 || t256 == 384 || t256 == 512 || (t256 >= 192 && t256 < 256))) return 51;  // K < 256 (the im2col input conv, one K tile): nothing for loaders to do, 128x160 is 15 us against 20  // 192: the 16x16 level's fused q/k/v projection (23 us against 26-33 for the other tiles)
    if (!n160 && conv && N % 128 == 0 && N >= 256 && t256 >= 512 && K >= 2304) return 53;  // K = 1152 (128 -> 256 at 256 x 256): 128x128 tiles, 0.42 vs 0.45 ms in-step
  }
  if ((tune & 1) && !conv && n160 && K >= 256 && (t256 == 256 || t256 == 512) && M % 256 == 0) return 11;
  if (t128 >= 512) return n160 ? 0 : 2;
  // Round 3 (late), measured INSIDE the step (TSD_GEMM_CFG_OVERRIDE + experiments/drivers/instep_sweep.sh; the repeated-launch microbenchmark
  // keeps the operands in the L2 and ranks these the other way round): dense GEMMs with exactly one 128-row tile per CU run the
  // staggered 128x160 tile with loader waves (54: 8192x640x640 19.4 -> 17.7 us, 8192x640x2560 42 -> 39.7 us), and the 256-tile
  // 64-row problems of the 16x16 level the 64x160 tile with loader waves (47: 2048x1280x1280 20.2 -> 19.2 us); TSD_GEMM_TUNE bit 3
  if ((tune & 8) && !conv && n160 && t128 == 256 && M % 128 == 0 && K < 5760) return 54;
  if ((tune & 8) && !conv && n160 && N >= 8192 && t128 >= 384) return 0;  // few rows, very wide (context K | V^T: 34 -> 29 us)
  if (t128 >= 192) return (K >= 2560 || !(tune & 2)) ? (n160 ? 5 : 8) : (n160 ? 1 : 3);
  if (K >= 5760) return n160 ? 6 : 9;
  {  // 256 tiles: measured on 2048x1280x1280; 128 tiles: the full-size UNet's 1024x1280x1280 at batch 4 (0.483 -> 0.463 ms for its 25 launches)
    const long long t64 = (long long)ceil_div(M, 64) * ceil_div(N, BN) * batch;
    if ((tune & 8) && !conv && n160 && (t64 == 256 || t64 == 128) && M % 64 == 0 && N >= 1280) return 47;
  }
  return n160 ? 7 : 10;
}

// rows / columns of one wave's output sub-tile for tile configuration `id` (FM*16, FN*16)
static void cfg_wave_tile(int id, int* bmw, int* bnw) {
  switch (id) {
    case 0: case 5: case 11: case 45: case 51: *bmw = 64; *bnw = 80; break;
    case 1: case 6: case 7: case 46: case 47: case 54: *bmw = 32; *bnw = 80; break;
    case 2: case 8: case 13: case 48: case 53: *bmw = 64; *bnw = 64; break;
    case 3: case 9: case 10: case 49: case 50: case 55: *bmw = 32; *bnw = 64; break;
    default: *bmw = 0; *bnw = 0; break;  // thin / experimental tiles: no epilogue statistics
  }
}
// EPI_GNSTATS geometry for a launch of this shape: slabs per sample (rows_per_sample / wave rows), or 0 when the tile
// it would run with cannot emit statistics for `groups` groups over N channels.
int gemm_gnstats_slabs(const tsd_ctx* ctx, int M, int N, int K, int batch, int conv, int rows_per_sample, int groups) {
  if (groups <= 0 || N % groups || (N & 7) || batch != 1) return 0;
  int bmw, bnw;
  cfg_wave_tile(choose_cfg(ctx->opt, M, N, K, batch, conv != 0, rows_per_sample), &bmw, &bnw);
  const int cpg = N / groups;
  if (!bmw || bnw % cpg || rows_per_sample % bmw || M % rows_per_sample) return 0;
  return rows_per_sample / 32;  // one slab per 32-row epilogue pass, independent of the tile shape
}

template <bool CONV>
static int dispatch(tsd_ctx* ctx, const GemmK& k, int batch) {
  const TsdOptions& o = ctx->opt;
  const int force_cfg = o.force_cfg;
  int id = force_cfg >= 0 ? force_cfg : (k.splitk > 1 ? k.sk_cfg : choose_cfg(o, k.M, k.N, k.K, batch, CONV));
  // halo-x measured: -3...5 % on the 128x128-tile convs of the VAE (N = 128 / 256 / 512), nothing on the 128x160 ones (TSD_CONV_HALO=2 turns those on too)
  // (the halo-x K order addresses W row-major: a launch that reads the K-tile-major weight copy keeps its plain tile)
  if (CONV && force_cfg < 0 && k.w_kts == 128u && hx_eligible(o, k) && (id == 2 || (id == 0 && o.conv_halo >= 2))) id += 30;
  // rounds 3-4: the fused-skip variant of the 128x128 two-blocks-per-CU tile kept its offset tables in scratch (48 B per lane) and the
  // decoder's 256 -> 128 residual block at 512 x 512 (K = 1152 + 256) ran its 64-row sibling instead (1.18 ms against 1.35).  Round 5:
  // the scratch is gone (conv_tap_ptrs); TSD_GEMM_SKIP128=0 restores the detour for A/B runs
  if (CONV && force_cfg < 0 && k.Cin1 > 0 && id == 2 && !o.skip128) id = 3;
  if ((id == 30 || id == 32) && !(CONV && hx_shape_ok(k) && k.w_kts == 128u)) TSD_FAIL(TSD_E_ARG, "gemm: halo-x tile configuration %d on an ineligible problem", id);
  return launch_by_id<CONV>(ctx, k, batch, id);
}

// The bench / check entries below pin the tile configuration through the context; the guard puts "dispatcher's choice" back on EVERY
// way out (a HIP_TRY / TSD_TRY early return used to leave later launches of the context pinned: ADVICE r04)
namespace {
struct ForceCfgGuard {
  tsd_ctx* ctx;
  ForceCfgGuard(tsd_ctx* c, int cfg) : ctx(c) { ctx->opt.force_cfg = cfg; }
  ~ForceCfgGuard() { ctx->opt.force_cfg = -1; }
  void set(int cfg) { ctx->opt.force_cfg = cfg; }
};
}  // namespace

// Debug/bench entry: time `iters` launches of one GEMM / conv3x3 problem on synthetic device data with a forced
// tile configuration (cfg < 0: the dispatcher's choice).  conv: M = B*Ho*Wo from (B,H,W,stride,ups), K = 9*Cin.
extern "C" int tsd_debug_gemm_bench(tsd_ctx* ctx, int conv, int B, int H, int W, int Cin, int N, int stride, int ups,
                                    int cfg, int iters, float* ms) {
  if (!ctx || !ms || iters <= 0) TSD_FAIL(TSD_E_ARG, "gemm_bench: bad argument");
  if (cfg >= N_GEMM_CFG) TSD_FAIL(TSD_E_ARG, "gemm_bench: cfg %d out of range", cfg);
  HIP_TRY(hipSetDevice(ctx->device));
  const int Hi = ups ? 2 * H : H, Wi = ups ? 2 * W : W;
  const int Ho = conv ? (Hi + 2 - 3) / stride + 1 : H, Wo = conv ? (Wi + 2 - 3) / stride + 1 : W;
  const int64_t M = (int64_t)B * Ho * Wo, K = conv ? 9 * (int64_t)Cin : Cin;
  // TSD_BENCH_WROT=R: R copies of the weights used round-robin, so each launch streams cold weights like a real step
  const int wrot = ctx->opt.bench_wrot;
  const int64_t na = (int64_t)B * H * W * Cin, nw1 = (int64_t)N * K, nw = nw1 * wrot, nc = M * N;
  TSD_TRY(ctx_reserve_arena(ctx, (size_t)(na + nw + 2 * nc) * 2 + (size_t)std::max(na, nw1) * 4 + (size_t)N * 4 + 8192));
  ctx->arena.top = 0;
  half_t* A = arena_alloc<half_t>(ctx, na);
  half_t* Wt = arena_alloc<half_t>(ctx, nw);
  half_t* C = arena_alloc<half_t>(ctx, nc);
  float* tmp = arena_alloc<float>(ctx, std::max(na, nw1));
  if (!A || !Wt || !C || !tmp) TSD_FAIL(TSD_E_ALLOC, "gemm_bench: arena");
  TSD_TRY(launch_fill_uniform(ctx, tmp, na, 1, 1, 1.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)std::min<int64_t>(na, 1 << 30), A, (int)std::min<int64_t>(na, 1 << 30), 1));
  for (int r = 0; r < wrot; r++) {
    TSD_TRY(launch_fill_uniform(ctx, tmp, nw1, 1, 2, 0.05f));
    TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)nw1, Wt + r * nw1, (int)nw1, 1));
  }
  GemmArgs g;
  g.A0 = A; g.lda0 = Cin; g.Wt = Wt; g.ldw = (int)K; g.M = (int)M; g.N = N; g.K = (int)K; g.C = C; g.ldc = N;
  if (conv) { g.conv = 1; g.Hs = H; g.Ws = W; g.Ho = Ho; g.Wo = Wo; g.Cin = Cin; g.stride = stride; g.pad = 1; g.ups = ups; }
  // TSD_BENCH_EPI: 0 plain store, 1 bias + residual (projection / conv2 epilogue), 2 bias + GEGLU
  const int epi_mode = ctx->opt.bench_epi;
  if (epi_mode) {
    float* bias = arena_alloc<float>(ctx, N);
    half_t* R = arena_alloc<half_t>(ctx, epi_mode == 1 ? nc : 0);
    if (!bias || (epi_mode == 1 && !R)) TSD_FAIL(TSD_E_ALLOC, "gemm_bench: arena");
    TSD_TRY(launch_fill_uniform(ctx, bias, N, 1, 3, 0.1f));
    g.bias = bias; g.epi = EPI_BIAS_N;
    if (epi_mode == 1) { HIP_TRY(hipMemsetAsync(R, 0, (size_t)nc * 2, ctx->stream)); g.R = R; g.ldr = N; g.epi |= EPI_RESIDUAL; }
    else { g.epi |= EPI_GEGLU; g.ldc = N / 2; }
  }
  ForceCfgGuard forced(ctx, cfg);
  int r = launch_gemm(ctx, g);
  if (r == TSD_OK) r = launch_gemm(ctx, g);
#ifdef TSD_GEMM_TS
  if (ctx->opt.gemm_ts) {
    const int nblk = 1 << 16;
    unsigned long long* dts = nullptr;
    HIP_TRY(hipMalloc((void**)&dts, (size_t)nblk * 64));
    HIP_TRY(hipMemset(dts, 0, (size_t)nblk * 64));
    g_ts = dts;
    r = launch_gemm(ctx, g);
    g_ts = nullptr;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> h((size_t)nblk * 8);
    HIP_TRY(hipMemcpy(h.data(), dts, (size_t)nblk * 64, hipMemcpyDeviceToHost));
    hipFree(dts);
    unsigned long long tmin = ~0ull, tmax = 0; int n = 0;
    for (int b = 0; b < nblk; b++) if (h[b * 8]) { n++; tmin = std::min(tmin, h[b * 8 + 5]); tmax = std::max(tmax, h[b * 8 + 6]); }
    fprintf(stderr, "[ts] blocks=%d span first-start..last-end = %.2f us (100 MHz realtime)\n", n, (tmax - tmin) * 0.01);
    {
      std::vector<double> st, en;
      for (int b = 0; b < nblk; b++) if (h[b * 8]) { st.push_back((h[b * 8 + 5] - tmin) * 0.01); en.push_back((h[b * 8 + 6] - tmin) * 0.01); }
      std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
      fprintf(stderr, "[ts] block start (us): med %.2f p90 %.2f max %.2f ; block end (us): min %.2f med %.2f max %.2f\n",
              st[st.size() / 2], st[st.size() * 9 / 10], st.back(), en.front(), en[en.size() / 2], en.back());
    }
    const char* nm[5] = {"start-after-first-block", "prologue", "main loop", "drain+barrier", "epilogue"};
    for (int ph = 1; ph < 5; ph++) {
      std::vector<unsigned long long> d;
      for (int b = 0; b < nblk; b++) if (h[b * 8]) d.push_back(h[b * 8 + ph] - h[b * 8 + ph - 1]);
      std::sort(d.begin(), d.end());
      fprintf(stderr, "[ts] %-24s min %8llu  med %8llu  p90 %8llu  max %8llu\n", nm[ph], d.front(), d[d.size() / 2], d[d.size() * 9 / 10], d.back());
    }
  }
#endif
  if (r != TSD_OK) return r;
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  const int alt = ctx->opt.bench_altcfg;  // alternate two kernels (cold I-cache probe)
  for (int i = 0; i < iters && r == TSD_OK; i++) {
    if (alt >= 0) forced.set((i & 1) ? alt : cfg);
    g.Wt = Wt + (int64_t)(i % wrot) * nw1;
    r = launch_gemm(ctx, g);
  }
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
  *ms = t / iters;
  ctx->arena.top = 0;
  return r;
}

// Debug entry: run one problem with tile configuration `cfg` and with the reference configuration `ref_cfg`, compare
// the two outputs on the host (the 2-stage configurations are validated against the oracle by the test-suite).
extern "C" int tsd_debug_gemm_check(tsd_ctx* ctx, int conv, int B, int H, int W, int Cin, int N, int stride, int ups,
                                    int cfg, int ref_cfg, float* max_abs_diff, float* max_abs_ref) {
  if (!ctx || !max_abs_diff || !max_abs_ref) TSD_FAIL(TSD_E_ARG, "gemm_check: bad argument");
  if (cfg >= N_GEMM_CFG || ref_cfg >= N_GEMM_CFG) TSD_FAIL(TSD_E_ARG, "gemm_check: cfg out of range");
  HIP_TRY(hipSetDevice(ctx->device));
  const int Hi = ups ? 2 * H : H, Wi = ups ? 2 * W : W;
  const int Ho = conv ? (Hi + 2 - 3) / stride + 1 : H, Wo = conv ? (Wi + 2 - 3) / stride + 1 : W;
  const int64_t M = (int64_t)B * Ho * Wo, K = conv ? 9 * (int64_t)Cin : Cin;
  const int64_t na = (int64_t)B * H * W * Cin, nw = (int64_t)N * K, nc = M * N;
  TSD_TRY(ctx_reserve_arena(ctx, (size_t)(na + nw + 2 * nc) * 2 + (size_t)std::max(na, nw) * 4 + 8192));
  ctx->arena.top = 0;
  half_t* A = arena_alloc<half_t>(ctx, na);
  half_t* Wt = arena_alloc<half_t>(ctx, nw);
  half_t* C0 = arena_alloc<half_t>(ctx, nc);
  half_t* C1 = arena_alloc<half_t>(ctx, nc);
  float* tmp = arena_alloc<float>(ctx, std::max(na, nw));
  if (!A || !Wt || !C0 || !C1 || !tmp) TSD_FAIL(TSD_E_ALLOC, "gemm_check: arena");
  TSD_TRY(launch_fill_uniform(ctx, tmp, na, 1, 1, 1.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)na, A, (int)na, 1));
  TSD_TRY(launch_fill_uniform(ctx, tmp, nw, 1, 2, 0.05f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)nw, Wt, (int)nw, 1));
  GemmArgs g;
  g.A0 = A; g.lda0 = Cin; g.Wt = Wt; g.ldw = (int)K; g.M = (int)M; g.N = N; g.K = (int)K; g.ldc = N;
  if (conv) { g.conv = 1; g.Hs = H; g.Ws = W; g.Ho = Ho; g.Wo = Wo; g.Cin = Cin; g.stride = stride; g.pad = 1; g.ups = ups; }
  int r;
  {
    ForceCfgGuard forced(ctx, ref_cfg);
    g.C = C0;
    r = launch_gemm(ctx, g);
    g.C = C1; forced.set(cfg);
    if (r == TSD_OK) r = launch_gemm(ctx, g);
  }
  if (r != TSD_OK) return r;
  std::vector<half_t> h0(nc), h1(nc);
  HIP_TRY(hipMemcpyAsync(h0.data(), C0, nc * 2, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(h1.data(), C1, nc * 2, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  float md = 0.f, mr = 0.f;
  for (int64_t i = 0; i < nc; i++) {
    const float a0 = (float)h0[i], a1 = (float)h1[i];
    const float d = fabsf(a0 - a1);
    if (!(d <= md)) md = d;  // NaN propagates
    if (fabsf(a0) > mr) mr = fabsf(a0);
  }
  *max_abs_diff = md; *max_abs_ref = mr;
  ctx->arena.top = 0;
  return TSD_OK;
}

extern "C" int tsd_debug_splitk_errors(tsd_ctx* ctx) {
  if (!ctx || !ctx->sk_flags) return 0;
  int v = -1;
  hipStreamSynchronize(ctx->stream);
  if (hipMemcpy(&v, ctx->sk_flags + 4095, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return v;
}

int launch_gemm(tsd_ctx* ctx, const GemmArgs& a) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) TSD_FAIL(TSD_E_SHAPE, "gemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  if (a.K % 64) TSD_FAIL(TSD_E_SHAPE, "gemm: K=%d must be a multiple of 64 (pad at pack time)", a.K);
  if (a.N % 4) TSD_FAIL(TSD_E_SHAPE, "gemm: N=%d must be a multiple of 4", a.N);
  if ((a.epi & EPI_GEGLU) && (a.N % 8)) TSD_FAIL(TSD_E_SHAPE, "gemm: GEGLU needs N %% 8 == 0");
  if (a.conv ? (a.epi & (EPI_GEGLU | EPI_BIAS_M)) : (a.epi & (EPI_ROWVEC | EPI_RES_UPS)))
    TSD_FAIL(TSD_E_ARG, "gemm: epilogue flags 0x%x are not available for %s", a.epi, a.conv ? "conv3x3" : "dense GEMMs");
  if (a.conv) {
    if (a.Cin % 64 || a.K != 9 * a.Cin + a.Cin1 + a.Cin2) TSD_FAIL(TSD_E_SHAPE, "conv3x3: Cin=%d K=%d", a.Cin, a.K);
    if (a.Cin1 || a.Cin2) {  // fused 1x1 skip (residual block): same resolution as the output, whole 64-channel chunks
      if (a.Cin1 <= 0 || a.Cin1 % 64 || a.Cin2 % 64 || !a.A1 || (a.Cin2 && !a.A2) || !a.Wt1 || a.stride != 1 || a.ups || a.Ho != a.Hs ||
          a.Wo != a.Ws || a.lda1 < a.Cin1 || (a.Cin2 && a.lda2 < a.Cin2) || a.ldw1 < a.Cin1 + a.Cin2)
        TSD_FAIL(TSD_E_ARG, "conv3x3: fused skip source (%d + %d channels) does not fit this convolution", a.Cin1, a.Cin2);
    }
    if (a.batch != 1) TSD_FAIL(TSD_E_ARG, "conv3x3: batch is folded into M");
  } else {
    if (a.K0 % 64) TSD_FAIL(TSD_E_SHAPE, "gemm: concat split K0=%d must be a multiple of 64", a.K0);
  }
  {  // the kernel addresses each operand slice through a 2 GiB buffer-descriptor window with 32-bit byte offsets
    const long long lim = 0x7ffffff0LL - 65536;
    const long long a_bytes = a.conv ? 2LL * (a.M / (a.Ho * a.Wo)) * a.Hs * a.Ws * a.lda0
                                     : 2LL * ((long long)(a.M - 1) * (a.lda0 > a.lda1 ? a.lda0 : a.lda1) + a.K);
    const long long w_bytes = a.w_kts ? (long long)(a.K / 64) * a.w_kts : 2LL * ((long long)(a.N - 1) * a.ldw + a.K);
    const long long s_bytes = a.conv && a.Cin1 ? 2LL * (a.M / (a.Ho * a.Wo)) * a.Hs * a.Ws * (a.lda1 > a.lda2 ? a.lda1 : a.lda2) : 0;
    if (a_bytes > lim || w_bytes > lim || s_bytes > lim)
      TSD_FAIL(TSD_E_SHAPE, "gemm: operand slice of %lld / %lld bytes exceeds the 2 GiB addressing window", a_bytes, w_bytes);
  }
  if (a.Vt) {
    const int BNt = (a.N % 160 == 0) ? 160 : 128;
    if (a.conv || a.batch != 1 || (a.epi & ~(EPI_BIAS_N)) || a.N % 8 || a.N <= 16 || a.vt_n0 <= 0 || a.vt_n0 % BNt || a.vt_S <= 0 || a.vt_S % 8 ||
        a.M % a.vt_S || a.vt_ld < a.vt_S || a.vt_ld % 8 || ctx->opt.force_cfg >= 0)
      TSD_FAIL(TSD_E_ARG, "gemm: transposed tail (n0 %d, rows per sample %d, pitch %d) does not fit this launch", a.vt_n0, a.vt_S, a.vt_ld);
  }
  if (a.epi & EPI_GNSTATS) {
    if ((a.epi & (EPI_GEGLU | EPI_OUT_F32)) || !a.gn_part ||
        a.gn_nslab != gemm_gnstats_slabs(ctx, a.M, a.N, a.K, a.batch, a.conv, a.gn_rows_per_sample, a.gn_groups) || a.gn_nslab <= 0)
      TSD_FAIL(TSD_E_ARG, "gemm: GroupNorm statistics requested for a shape/tile that cannot emit them");
  }
  // split-K workspace (arena: the planning pass sees the same allocation) and the per-context arrival flags
  float* sk_ws = nullptr;
  int sk_cfg = 0;
  const int ways = ctx->opt.force_cfg < 0 ? splitk_plan(ctx->opt, a.M, a.N, a.K, a.batch, a.rows_per_sample_hint, &sk_cfg) : 1;
  const bool splitk = ways > 1;
  if (splitk) {
    const int BN = (a.N % 160 == 0) ? 160 : 128, BM = (sk_cfg == 7 || sk_cfg == 10 || sk_cfg == 6 || sk_cfg == 9) ? 64 : (sk_cfg == 51 ? 256 : 128);
    sk_ws = arena_alloc<float>(ctx, (int64_t)(ways - 1) * ceil_div(a.M, BM) * ceil_div(a.N, BN) * BM * BN);
    if (!sk_ws) TSD_FAIL(TSD_E_ALLOC, "gemm: split-K workspace exhausted");
  }
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, a.conv ? KC_CONV : KC_GEMM, a.M, a.N, a.K, a.batch);
  GemmK k;
  k.A0 = a.A0; k.A1 = a.A1; k.Wt = a.Wt; k.R = a.R; k.zeros = ctx->zeros;
  k.bias = a.bias; k.rowvec = a.rowvec; k.C = a.C;
  k.sA = a.sA; k.sW = a.sW; k.sC = a.sC; k.sR = a.sR;
  k.lda0 = a.lda0; k.lda1 = a.lda1; k.K0 = (a.A1 ? a.K0 : a.K); k.ldw = a.ldw; k.ldr = a.ldr; k.ldc = a.ldc;
  k.rowvec_ld = a.rowvec_ld; k.rows_per_batch = a.rows_per_batch > 0 ? a.rows_per_batch : 1;
  k.M = a.M; k.N = a.N; k.K = a.K;
  k.Hs = a.Hs; k.Ws = a.Ws; k.Ho = a.Ho; k.Wo = a.Wo; k.Cin = a.Cin; k.stride = a.stride; k.pad = a.pad; k.ups = a.ups;
  k.A2 = a.A2; k.Wt1 = a.Wt1; k.lda2 = a.lda2; k.Cin1 = a.conv ? a.Cin1 : 0; k.Cin2 = a.conv ? a.Cin2 : 0; k.ldw1 = a.ldw1;
  k.epi = a.epi; k.tiles_n = 0; k.out_scale = a.out_scale;
  k.w_kts = a.w_kts ? (unsigned)a.w_kts : 128u;
  if (a.w_kts && (a.ldw != 64 || a.w_kts < a.N * 128)) TSD_FAIL(TSD_E_ARG, "gemm: K-tile-major W needs ldw = 64 and a tile stride >= N * 128");
  k.gn_part = a.gn_part; k.gn_cpg = a.gn_groups > 0 ? a.N / a.gn_groups : 1; k.gn_G = a.gn_groups; k.gn_hw = a.gn_rows_per_sample;
  k.gn_nslab = a.gn_nslab;
  k.vt = a.Vt; k.vt_sB = a.vt_sB; k.vt_n0 = a.vt_n0; k.vt_ld = a.vt_ld; k.vt_S = a.vt_S;
  k.splitk = splitk ? ways : 1; k.sk_cfg = sk_cfg; k.sk_ws = sk_ws; k.sk_flags = nullptr; k.sk_epoch = 0;
  if (splitk) {
    if (!ctx->sk_flags) {
      HIP_TRY(hipMalloc((void**)&ctx->sk_flags, 4096 * sizeof(int)));
      HIP_TRY(hipMemsetAsync(ctx->sk_flags, 0, 4096 * sizeof(int), ctx->stream));
    }
    k.sk_flags = ctx->sk_flags;
    if (++ctx->sk_epoch == 0) ctx->sk_epoch = 1;
    k.sk_epoch = (int)ctx->sk_epoch;
  }
  return a.conv ? dispatch<true>(ctx, k, a.batch) : dispatch<false>(ctx, k, a.batch);
}

// ---- what this board sustains: register-resident dense fp16 MFMA loop (no LDS, no memory) ------------------------------
// 4 waves per SIMD, each issuing independent v_mfma_f32_16x16x32_f16 back to back on pseudo-random operands, for about
// `ms_target` milliseconds.  Reports the achieved TFLOP/s and the shader clock during the run (s_memtime ticks against the
// 100 MHz s_memrealtime).  The dense peak of MI355X_MICROARCH.md (2.5 PF) assumes the 2.4 GHz boost clock; under a
// matrix-pipe load the board's power limit sets the clock, and this is the ceiling the GEMM-class kernels can be priced
// against in practice (bench.py reports it next to, never instead of, the nominal peak).
__device__ unsigned long long g_probe_ts[4];
__global__ __launch_bounds__(256) void mfma_probe_kernel(float* sink, int iters, unsigned seed) {
  typedef _Float16 ph8 __attribute__((ext_vector_type(8)));
  typedef float pf4 __attribute__((ext_vector_type(4)));
  ph8 a[2], b;
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; a[i][e] = (_Float16)(((int)(s >> 20) - 2048) * (1.f / 2048.f)); }
#pragma unroll
  for (int e = 0; e < 8; e++) { s = s * 1664525u + 1013904223u; b[e] = (_Float16)(((int)(s >> 20) - 2048) * (1.f / 2048.f)); }
  pf4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_probe_ts[0] = __builtin_amdgcn_s_memtime(); g_probe_ts[2] = __builtin_amdgcn_s_memrealtime(); }
  for (int it = 0; it < iters; it += 4) {  // 32 MFMAs per trip on two accumulators; the 4 waves of a SIMD interleave
#pragma unroll
    for (int i = 0; i < 16; i++) {
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1], b, acc1, 0, 0, 0);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) { g_probe_ts[1] = __builtin_amdgcn_s_memtime(); g_probe_ts[3] = __builtin_amdgcn_s_memrealtime(); }
  float r = acc0[0] + acc0[1] + acc0[2] + acc0[3] + acc1[0] + acc1[1] + acc1[2] + acc1[3];
  if (r == 1234.5678f) sink[threadIdx.x] = r;  // keeps the loop alive; practically never true
}

extern "C" int tsd_debug_mfma_sustained(tsd_ctx* ctx, float ms_target, float* tflops, float* clock_ghz) {
  if (!ctx || !tflops || !clock_ghz || !(ms_target > 0.f) || ms_target > 1000.f) TSD_FAIL(TSD_E_ARG, "mfma_sustained: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
  const int blocks = prop.multiProcessorCount * 4;  // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  TSD_TRY(ctx_reserve_arena(ctx, 4096));
  float* sink = (float*)ctx->arena.base;
  int iters = 2000;  // x 8 MFMAs per wave
  float ms = 0.f;
  for (int pass = 0; pass < 3; pass++) {  // pass 0 calibrates, pass 1 warms the board up to its power state, pass 2 is reported
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(mfma_probe_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sink, iters, 12345u + pass);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    if (pass == 0) { iters = (int)std::min(50.0e6, std::max(100.0, iters * (double)ms_target / std::max(ms, 1e-3f))); iters = (iters + 3) & ~3; }
  }
  unsigned long long ts[4];
  HIP_TRY(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_probe_ts), sizeof(ts)));
  const double flop = (double)blocks * 4 /*waves*/ * (double)iters * 8 * (2.0 * 16 * 16 * 32);
  *tflops = (float)(flop / (ms * 1e-3) / 1e12);
  const double dr = (double)(ts[3] - ts[2]) * 10.0;  // ns
  *clock_ghz = dr > 0 ? (float)((double)(ts[1] - ts[0]) / dr) : 0.f;
  return TSD_OK;
}
