// kernels_chain.hip - the row-local tail of `Unet_Attention_Block.forward` as ONE kernel (C = 320: the 64x64 level).
//
// After self-attention everything in the block is local to a token row (diffusion.mojo:125-146):
//   tok2 = ao . Wso^T + b + tok                                  self-attention out_proj + residual      (:125-126, attention.mojo:64)
//   tok3 = CrossAttn(LN(tok2), context) . Wco^T + b + tok2       q_proj, 77-key attention, out_proj      (:128-133, attention.mojo:96-118)
//   tok4 = (a * gelu(g)) . W2^T + b2 + tok3,  (a, g) = LN(tok3) . W1^T + b1   GEGLU feed-forward         (:135-143)
//   out  = tok4 . Wout^T + b + x                                 1x1 conv + long residual                (:146)
// The unfused graph runs these as 9 launches that move the 21 MB token tensor through HBM ~30 times (and the 84 MB GEGLU
// intermediate twice).  Here a workgroup owns 64 token rows for the whole chain:
//   * the residual stream T[64][320] lives in REGISTERS (fp32, in the MFMA accumulator layout: the per-column epilogue
//     terms and the residual adds are lane-local);
//   * the A operand of every GEMM (ao, LN(T), q, attention output, GEGLU activations, tok4) is an fp16 tile in LDS in the
//     swizzled 128-B-row layout the MFMA fragment reads want, written straight from the accumulator registers;
//   * the weights of all stages are PRE-PACKED (launch_attn_tail_pack, once per model) into one stream of [rows][32 k]
//     tiles that are byte images of their LDS layout (row permutation and bank swizzle baked in): staging a tile is a
//     linear 20 KB buffer_load ... lds copy with no per-lane address math, and the stream runs four tiles ahead of the
//     MFMAs through a 5-slot LDS ring, across stage boundaries, behind counted s_waitcnt vmcnt;
//   * the 77 context keys / values of the sample (projected once per step) are staged in the ring region; the scores
//     never leave registers: swapped-operand QK^T leaves each lane with the scores of ONE query row, and the key -> MFMA
//     row permutation makes the fp16 probabilities the A fragments of the P.V MFMAs without any data movement;
//   * LayerNorm statistics are an in-lane sum + two lane shuffles + one LDS exchange between the two column waves
//     (per-wave mean / M2 merged exactly);
//   * the output's GroupNorm statistics (for the next residual block) come from the rounded values in registers.
// MFMA: v_mfma_f32_16x16x32_f16 with swapped operands (D = Wfrag x Afrag^T), 4 waves as 2(M) x 2(N), a wave tile is
// 32 rows x 160 columns (FM = 2, FN = 10), one wave per SIMD.
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

#ifndef TSD_CHAIN_PIPE
#define TSD_CHAIN_PIPE 1  // 0 = the round-2 per-chunk GEGLU loop (A/B builds)
#endif

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct TailK {
  const half_t* ao; const half_t* tok; const half_t* x; half_t* out;
  int ld_ao, ld_tok, ld_x, ld_out;
  const half_t* wstream;                      // launch_attn_tail_pack() image of Wso, Wq, Wco, W1, W2, Wout
  const float *bso, *bco, *b1, *b2, *bout;
  const half_t* Kc; int ldk; long long sK;    // projected context keys   [B][>=T rows][ldk], this block's columns
  const half_t* Vt; int ldvt; long long sVt;  // projected context values [B][C rows][ldvt]  (key-contiguous)
  int T;                                      // valid keys (<= 80)
  int M, S;                                   // token rows, rows per sample (S % 64 == 0)
  float qscale;                               // softmax scale * log2(e), folded into q
  float eps;
  float* gn_part; int gn_nslab;               // output GroupNorm(32) statistics [B][nslab][32][2] (nullptr: none)
  // head kernel only (KIND_HEAD): GroupNorm statistics of x, conv_in bias, q/k and V^T outputs (tok is an OUTPUT there)
  const float* gn_stats;                      // [B][32][2] (mean, 1 / (sigma + eps)) of the block input
  const float* b_in;
  half_t* tok_out; half_t* qk; int ld_qk; half_t* vt; int ld_vt; long long s_vt;
};

namespace {
constexpr int C = 320, BM = 64;
constexpr int A_OFF = 0, A_KT = 8192;                 // A tile: 5 k-tiles x [64 rows][128 B]
constexpr int ACT_OFF = 40960;                        // GEGLU activations: 2 k-tiles x [64][128 B]
constexpr int RING_OFF = ACT_OFF + 16384, SLOT = 20480, NSLOT = 5;  // ring of [2 halves][160 rows][64 B] weight tiles
constexpr int SCR_OFF = RING_OFF + NSLOT * SLOT;      // LayerNorm exchange: [64 rows][2 column waves] x (mean, M2)
constexpr int LDS_BYTES = SCR_OFF + 1024;
static_assert(LDS_BYTES <= 163840, "LDS budget");
// the weight stream (bytes): tiles of 32 k.  Full tiles hold 320 rows (20480 B), GEGLU-1 tiles 256 rows (16384 B).
constexpr int TILE_FULL = 20480, TILE_G1 = 16384;
constexpr int SEG0_TILES = 20;                                    // Wso (10), Wq (10)
constexpr int SEG0_BYTES = SEG0_TILES * TILE_FULL;
constexpr int FFN_CHUNK_BYTES = 10 * TILE_G1 + 4 * TILE_FULL;     // GEGLU-1 chunk (K = 320), GEGLU-2 chunk (K = 128)
constexpr int SEG1_TILES = 10 + 10 * 14 + 10;                     // Wco, 10 x (W1 chunk, W2 chunk), Wout
constexpr int SEG1_BYTES = 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES + 10 * TILE_FULL;
constexpr int STREAM_BYTES = SEG0_BYTES + SEG1_BYTES;
constexpr int KIND_TAIL = 0, KIND_HEAD = 1;
constexpr int HEAD_TILES = 40;                                    // conv_in (10), Wq (10), Wk (10), Wv (10)
constexpr int HEAD_STREAM_BYTES = HEAD_TILES * TILE_FULL;

__device__ __forceinline__ float gelu_tanh_c(float x) {  // helpers/utils.mojo:1914 (see kernels_gemm.hip)
  const float c2 = -2.f * 0.7978845608028654f * 1.4426950408889634f;
  const float t = __builtin_amdgcn_exp2f(c2 * (x + 0.044715f * x * x * x));
  return x * __builtin_amdgcn_rcpf(1.f + t);
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// bank swizzle of the 64-B weight rows: 16-B chunk c of LDS row rho sits at chunk c ^ wswz(rho) (conflict-free
// ds_read_b128 across the instruction's four 16-lane service groups)
__host__ __device__ __forceinline__ int wswz(int rho) { return 3 * ((rho >> 2) & 1); }
}  // namespace

// ---- weight pre-packing: reference-layout fp16 matrices -> the kernel's tile stream --------------------------------
// One thread per 16-B chunk of the stream.  Tile t of a matrix W[N][ldw] (k columns 32t .. 32t+31, offset kofs):
//   image byte  half*HB + rho*64 + pc*16  <-  W[n(half, rho)][kofs + 32t + 8*(pc ^ wswz(rho)) .. +8]
// with n(half, rho) = half*NH + (ii>>2)*(NH/4) + fn*4 + (ii&3), fn = rho>>4, ii = rho&15 (NH = rows per half): output
// lane (g, r) of fragment b then owns column half*NH + g*(NH/4) + b*4 + r - NH/4 consecutive columns per lane.
struct PackSrc { const half_t* w; int ldw; };
__global__ void k_attn_tail_pack(PackSrc so, PackSrc q, PackSrc co, PackSrc w1, PackSrc w2, PackSrc wo, half_t* dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;  // 16-B chunk index in the stream
  if (ci >= STREAM_BYTES / 16) return;
  int byte = ci * 16;
  const half_t* W; int ldw, row0 = 0, kofs = 0, NH, t;
  if (byte < SEG0_BYTES) {
    const int ti = byte / TILE_FULL; byte -= ti * TILE_FULL;
    W = ti < 10 ? so.w : q.w; ldw = ti < 10 ? so.ldw : q.ldw; t = ti % 10; NH = 160;
  } else {
    byte -= SEG0_BYTES;
    if (byte < 10 * TILE_FULL) { t = byte / TILE_FULL; byte -= t * TILE_FULL; W = co.w; ldw = co.ldw; NH = 160; }
    else if (byte < 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES) {
      byte -= 10 * TILE_FULL;
      const int jc = byte / FFN_CHUNK_BYTES; byte -= jc * FFN_CHUNK_BYTES;
      if (byte < 10 * TILE_G1) { t = byte / TILE_G1; byte -= t * TILE_G1; W = w1.w; ldw = w1.ldw; row0 = jc * 256; NH = 128; }
      else { byte -= 10 * TILE_G1; t = byte / TILE_FULL; byte -= t * TILE_FULL; W = w2.w; ldw = w2.ldw; kofs = jc * 128; NH = 160; }
    } else {
      byte -= 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES;
      t = byte / TILE_FULL; byte -= t * TILE_FULL; W = wo.w; ldw = wo.ldw; NH = 160;
    }
  }
  const int half = byte / (NH * 64), rb = byte - half * NH * 64;
  const int rho = rb >> 6, pc = (rb >> 4) & 3;
  const int fn = rho >> 4, ii = rho & 15;
  const int n = row0 + half * NH + (ii >> 2) * (NH / 4) + fn * 4 + (ii & 3);
  const int k = kofs + 32 * t + 8 * (pc ^ wswz(rho));
  *(h8*)(dst + (size_t)ci * 8) = *(const h8*)(W + (size_t)n * ldw + k);
}

// head stream: conv_in, then the q / k / v row blocks of in_proj (rows 0..319 / 320..639 / 640..959), all K = 320.
// The v tiles keep the IDENTITY row order (LDS row rho of half h = weight row h*160 + rho): that stage runs with swapped
// MFMA operands and wants consecutive channels across a fragment's 16 lanes.
__global__ void k_attn_head_pack(PackSrc cin, PackSrc inproj, half_t* dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= HEAD_STREAM_BYTES / 16) return;
  int byte = ci * 16;
  const int ti = byte / TILE_FULL; byte -= ti * TILE_FULL;
  const int mat = ti / 10, t = ti - mat * 10;
  const half_t* W = mat == 0 ? cin.w : inproj.w;
  const int ldw = mat == 0 ? cin.ldw : inproj.ldw, row0 = mat == 0 ? 0 : (mat - 1) * 320;
  const int half = byte / 10240, rb = byte - half * 10240;
  const int rho = rb >> 6, pc = (rb >> 4) & 3, fn = rho >> 4, ii = rho & 15;
  const int n = row0 + half * 160 + (mat == 3 ? rho : (ii >> 2) * 40 + fn * 4 + (ii & 3));
  const int k = 32 * t + 8 * (pc ^ wswz(rho));
  *(h8*)(dst + (size_t)ci * 8) = *(const h8*)(W + (size_t)n * ldw + k);
}

#ifdef TSD_CHAIN_TS
__device__ unsigned long long g_chain_ts[2 * 1024 * 16];  // per kernel kind and block: s_memtime at the phase marks (experiment build only)
#define CTS(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_chain_ts[(KIND * 1024 + blockIdx.x) * 16 + (i)] = (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CTS(i) do { } while (0)
#endif

template <int KIND>
__global__ __launch_bounds__(256, 1) void attn_chain_kernel(const TailK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int rsel = lane & 15, key = lane & 7, g = lane >> 4;
  const int lrow = lane >> 3, cch = (lane & 7) ^ lrow;  // gather DMA: LDS row within an 8-row piece, logical 16-B chunk fetched
  const int m0 = blockIdx.x * BM;
  const int bsmp = m0 / p.S;
  CTS(0);
  CTS(14);

  // ---- the weight stream ---------------------------------------------------------------------------------------
  // byte offset / size class of tile gi of a segment (wave-uniform)
  auto tile_off = [&](int seg, int gi, bool& g1, bool& live) {
    g1 = false; live = true;
    if (KIND == KIND_HEAD) { live = gi < HEAD_TILES; return gi * TILE_FULL; }
    if (seg == 0) { live = gi < SEG0_TILES; return gi * TILE_FULL; }
    if (gi < 10) return SEG0_BYTES + gi * TILE_FULL;
    if (gi < 150) {
#if TSD_CHAIN_PIPE
      // visiting order W1(0) | W1(1) W2(0) | W1(2) W2(1) | ... | W1(9) W2(8) | W2(9): chunk j's activation is computed under the
      // MFMAs of chunk j-1's second GEMM (the packed image keeps its [W1(j), W2(j)] layout - only the walk changes)
      const int q = gi - 10;
      int jc, r; bool second;
      if (q < 10) { jc = 0; r = q; second = false; }
      else if (q < 136) { const int pq = (q - 10) / 14, pr = (q - 10) - 14 * pq; second = pr >= 10; jc = second ? pq : pq + 1; r = second ? pr - 10 : pr; }
      else { jc = 9; r = q - 136; second = true; }
      const int base = SEG0_BYTES + 10 * TILE_FULL + jc * FFN_CHUNK_BYTES;
      if (!second) { g1 = true; return base + r * TILE_G1; }
      return base + 10 * TILE_G1 + r * TILE_FULL;
#else
      const int q = gi - 10, jc = q / 14, r = q - 14 * jc;
      const int base = SEG0_BYTES + 10 * TILE_FULL + jc * FFN_CHUNK_BYTES;
      if (r < 10) { g1 = true; return base + r * TILE_G1; }
      return base + 10 * TILE_G1 + (r - 10) * TILE_FULL;
#endif
    }
    live = gi < SEG1_TILES;
    return SEG0_BYTES + 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES + (gi - 150) * TILE_FULL;
  };
#ifdef TSD_CHAIN_NODMA  // ablation build: zero-size descriptor = no weight traffic (results are wrong, the timing is the point)
  const rsrc_t rw = make_rsrc(p.wstream, 0), rdead = make_rsrc(p.wstream, 0);
#else
  const rsrc_t rw = make_rsrc(p.wstream, KIND == KIND_HEAD ? HEAD_STREAM_BYTES : STREAM_BYTES), rdead = make_rsrc(p.wstream, 0);
#endif
  const unsigned lane16 = lane * 16;
  // piece i (of 5 per wave) of tile gi -> ring slot: 1-KiB wave-instruction number wave + 4*i of the tile image
  auto piece = [&](int off, bool g1, bool live, int slot, int i) {
    const int j = wave + 4 * i;
    const bool on = live && !(g1 && j >= 16);
#ifdef TSD_CHAIN_NOPIECE  // ablation build: no DMA instructions at all after the first tiles of segment 1 (timing only)
    if (off >= SEG0_BYTES + 12 * TILE_FULL) return;
#endif
    blds16(on ? rw : rdead, lane16, (unsigned)(off + j * 1024), smem + RING_OFF + slot * SLOT + j * 1024);
  };
  int gt = 0, sl = 0;  // tile counter within the segment ; ring slot of tile gt
  auto seg_begin = [&](int seg) {
    gt = 0; sl = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
      bool g1, live;
      const int off = tile_off(seg, t, g1, live);
#pragma unroll
      for (int i = 0; i < 5; i++) if (!(g1 && i == 4)) piece(off, g1, live, t, i);
    }
  };

  // ---- one GEMM stage: acc[2][FN] (+)= Atile[64][nk*32] . W^T over the next nk tiles of the stream -----------------
  // EXTRA = VMEM loads the caller issued since the last DMA piece (bias / residual prefetch): they are younger than the
  // tiles the first wait is for.
  const int a_rd = (wm * 32 + rsel) * 128;
  const int w_rd = rsel * 64 + ((g ^ wswz(rsel)) << 4);
  auto gemm = [&](auto fn_c, auto seg_c, auto zero_c, auto extra_c, f4 (&acc)[2][10], int a_base, int nk, auto swap_c) {
    constexpr int FN = decltype(fn_c)::value, SEG = decltype(seg_c)::value, EXTRA = decltype(extra_c)::value;
    constexpr bool ZERO = decltype(zero_c)::value;
    // SWAP: D = Afrag x Wfrag^T instead of Wfrag x Afrag^T - a lane then holds ONE weight row (= lane & 15 of fragment b)
    // and FOUR consecutive token rows (4g + r of fragment a): the transposed output the V^T projection needs
    constexpr bool SWAP = decltype(swap_c)::value;
    if (ZERO) {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // Software pipeline over the tiles (one wave per SIMD: nobody else hides the LDS latency): the fragment reads of
    // tile t are issued right after its barrier and land while the MFMAs of tile t-1 run from the other register set.
    h8 afA[2], wfA[FN], afB[2], wfB[FN];
    // One pipeline step = barrier of tile gt, then the 2*FN MFMAs of tile gt-1 (register set p*) with the 2 + FN fragment
    // reads of tile gt (into set n*) and the wave's DMA instructions for tile gt+4 spread between them: a burst of 48
    // ds_read_b128 from the four waves right after the barrier would hold the LDS (and every wave's issue) for ~200 cycles.
    auto step = [&](int kt, h8 (&af)[2], h8 (&wf)[FN], const h8 (&paf)[2], const h8 (&pwf)[FN], bool have_prev, bool have_next) {
      bool g1 = false, live = false;
      int off = 0, s4 = 0;
      const char *sA = smem, *sW = smem;
      if (have_next) {
        // tile gt has landed: the three younger tiles (and the caller's EXTRA loads, which sit between tile gt+3 and tile
        // gt+4 of the stage's first tile in the queue) may stay in flight
        int younger = EXTRA > 0 && kt < 4 ? EXTRA : 0;
#pragma unroll
        for (int d = 1; d <= 3; d++) { bool yg1, ylive; tile_off(SEG, gt + d, yg1, ylive); younger += yg1 ? 4 : 5; }
        switch (younger - (EXTRA > 0 && kt < 4 ? EXTRA : 0)) {
          case 12: if (EXTRA > 0 && kt < 4) wait_vm<12 + EXTRA>(); else wait_vm<12>(); break;
          case 13: if (EXTRA > 0 && kt < 4) wait_vm<13 + EXTRA>(); else wait_vm<13>(); break;
          case 14: if (EXTRA > 0 && kt < 4) wait_vm<14 + EXTRA>(); else wait_vm<14>(); break;
          default: if (EXTRA > 0 && kt < 4) wait_vm<15 + EXTRA>(); else wait_vm<15>(); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile gt-1 are complete: its slot may be refilled
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sA = smem + a_base + (kt >> 1) * A_KT + a_rd + ((((kt & 1) * 4 + g) ^ key) << 4);
        sW = smem + RING_OFF + sl * SLOT + wn * (FN == 10 ? 10240 : 8192) + w_rd;
        off = tile_off(SEG, gt + 4, g1, live);
        s4 = sl == 0 ? 4 : sl - 1;  // (sl + 4) % 5: the slot tile gt-1 just left
      }
      constexpr int NM = 2 * FN, NR = 2 + FN;
      if (!have_prev) {  // first tile of the stage: nothing to multiply yet
#pragma unroll
        for (int a = 0; a < 2; a++) af[a] = *(const h8*)(sA + a * 2048);
#pragma unroll
        for (int b = 0; b < FN; b++) wf[b] = *(const h8*)(sW + b * 1024);
#pragma unroll
        for (int i = 0; i < 5; i++) if (!(g1 && i == 4)) piece(off, g1, live, s4, i);
      } else {
        int ri = 0, pi = 0;
#pragma unroll
        for (int q = 0; q < NM; q++) {
          const int b = q >> 1, a = q & 1;
          if (SWAP) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(paf[a], pwf[b], acc[a][b], 0, 0, 0);
          else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[b], paf[a], acc[a][b], 0, 0, 0);
          if (have_next) {
            __builtin_amdgcn_sched_barrier(0);
            // reads: one after each of the first NR MFMAs ... (A fragments first: every MFMA of the next step needs one)
            if (q < NR) {
              if (q < 2) af[q] = *(const h8*)(sA + q * 2048);
              else wf[q - 2] = *(const h8*)(sW + (q - 2) * 1024);
            }
            // ... DMA: one after every third / fourth MFMA
            if ((q + 1) % (NM / 5) == 0 && (q + 1) / (NM / 5) - 1 < 5) {
              const int i = (q + 1) / (NM / 5) - 1;
              if (!(g1 && i == 4)) piece(off, g1, live, s4, i);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (have_next) {
        gt++;
        sl = sl == 4 ? 0 : sl + 1;
      }
    };
    for (int kt = 0; kt < nk; kt += 2) {  // nk is even
      step(kt, afA, wfA, afB, wfB, kt > 0, true);
      step(kt + 1, afB, wfB, afA, wfA, true, true);
    }
    step(nk, afA, wfA, afB, wfB, true, false);
  };
  using I8 = std::integral_constant<int, 8>;
  using I10 = std::integral_constant<int, 10>;
  using I20 = std::integral_constant<int, 20>;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using Yes = std::true_type;
  using No = std::false_type;

  // ---- accumulator-layout helpers: lane (rsel, g) of fragment row a holds columns cbase + b*4 + r ------------------
  const int cbase = wn * 160 + g * 40;
  auto load_cols = [&](const float* vec, f4 (&bv)[10]) {
#pragma unroll
    for (int b = 0; b < 10; b++) bv[b] = *(const f4*)(vec + cbase + b * 4);
  };
  // fp16 rows (residual sources): 40 consecutive columns = five 16-B loads per fragment row
  auto load_rows_raw = [&](const half_t* src, int ld, h8 (&raw)[2][5]) {
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const half_t* rp = src + (long long)(m0 + wm * 32 + a * 16 + rsel) * ld + cbase;
#pragma unroll
      for (int q = 0; q < 5; q++) raw[a][q] = *(const h8*)(rp + q * 8);
    }
  };
  // write fp16((v - sub) * mul) into the A tile (swizzled A-operand layout): global 16-B chunk index = wn*20 + g*5 + q
  auto store_a_tile = [&](const f4 (&v)[2][10], float mul0, float sub0, float mul1, float sub1) {
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int row = wm * 32 + a * 16 + rsel;
      const float mul = a ? mul1 : mul0, sub = a ? sub1 : sub0;
#pragma unroll
      for (int q = 0; q < 5; q++) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] = (half_t)((v[a][2 * q + (j >> 2)][j & 3] - sub) * mul);
        const int cg = wn * 20 + g * 5 + q;
        *(h8*)(smem + A_OFF + (cg >> 3) * A_KT + row * 128 + (((cg & 7) ^ key) << 4)) = o;
      }
    }
  };
  // LayerNorm of the residual stream into the A tile: (x - mean) / (sigma + eps), population sigma, no affine
  // (helpers/utils.mojo:2052-2061 via :1845-1885; App.A D8).  Each column wave computes the exact two-pass mean / M2
  // of its 160 columns; the two are merged after ONE exchange (Chan et al.).
  float* scr = (float*)(smem + SCR_OFF);
  auto layernorm_to_a = [&](const f4 (&v)[2][10]) {
    float mw[2], m2w[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < 10; b++) t += (v[a][b][0] + v[a][b][1]) + (v[a][b][2] + v[a][b][3]);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      mw[a] = t * (1.f / 160.f);
      float u = 0.f;
#pragma unroll
      for (int b = 0; b < 10; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const float d = v[a][b][r] - mw[a]; u += d * d; }
      u += __shfl_xor(u, 16);
      u += __shfl_xor(u, 32);
      m2w[a] = u;
      if (g == 0) *(f2*)(scr + ((wm * 32 + a * 16 + rsel) * 2 + wn) * 2) = f2{mw[a], u};
    }
    lds_barrier();  // also: every wave is past its last read of the old A tile
    float mean[2], rs[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const f2 o = *(const f2*)(scr + ((wm * 32 + a * 16 + rsel) * 2 + (wn ^ 1)) * 2);
      const float dm = mw[a] - o[0];
      mean[a] = 0.5f * (mw[a] + o[0]);
      const float m2 = m2w[a] + o[1] + dm * dm * 80.f;
      rs[a] = 1.f / (sqrtf(m2 * (1.f / C)) + p.eps);
    }
    store_a_tile(v, rs[0], mean[0], rs[1], mean[1]);
  };

  if constexpr (KIND == KIND_HEAD) {
    // =============================================================================================================
    // head of the block (diffusion.mojo:116-124): GroupNorm-apply -> conv_in (1x1) -> tok ; LayerNorm -> q, k, V^T
    f4 T[2][10], acc[2][10], accq[2][10], bv[10];
    h8 raw[2][5];
    load_rows_raw(p.x, p.ld_x, raw);
    f2 gst[4];  // (mean, 1/(sigma+eps)) of this lane's four groups of 10 channels
#pragma unroll
    for (int k = 0; k < 4; k++) gst[k] = *(const f2*)(p.gn_stats + ((long long)bsmp * 32 + wn * 16 + g * 4 + k) * 2);
    load_cols(p.b_in, bv);
    seg_begin(0);
    // GroupNorm (32 groups, eps 1e-6, no SiLU; helpers/utils.mojo:1845-1885) of the x tile -> A tile
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int row = wm * 32 + a * 16 + rsel;
#pragma unroll
      for (int q = 0; q < 5; q++) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int grp = (q * 8 + j) / 10;
          o[j] = (half_t)(((float)raw[a][q][j] - gst[grp][0]) * gst[grp][1]);
        }
        const int cg = wn * 20 + g * 5 + q;
        *(h8*)(smem + A_OFF + (cg >> 3) * A_KT + row * 128 + (((cg & 7) ^ key) << 4)) = o;
      }
    }
    CTS(1);
    // ---- tok = GN(x) . Wc^T + b  (diffusion.mojo:117) ; the first tile barrier publishes the A tile -----------------------
    // No global store happens before the last GEMM is done: stores share vmcnt with the DMA stream, and draining them
    // (an HBM write round trip) in the middle of the kernel costs more than any stage.  tok / q / k wait in registers.
    gemm(I10{}, I0{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
    h8 tokh[2][5];
#pragma unroll
    for (int a = 0; a < 2; a++) {
#pragma unroll
      for (int b = 0; b < 10; b++) T[a][b] = acc[a][b] + bv[b];
#pragma unroll
      for (int q = 0; q < 5; q++)
#pragma unroll
        for (int j = 0; j < 8; j++) tokh[a][q][j] = (half_t)T[a][2 * q + (j >> 2)][j & 3];
    }
    CTS(2);
    layernorm_to_a(T);   // the LayerNorm sees the fp32 tok (the unfused graph normalises its fp16 rounding)
    CTS(3);
    // ---- q, k = LN(tok) . W^T  (helpers/attention.mojo:29, in_bias = False) -----------------------------------------------
    f4 acck[2][10];
    gemm(I10{}, I0{}, Yes{}, I0{}, accq, A_OFF, 10, No{});
    gemm(I10{}, I0{}, Yes{}, I0{}, acck, A_OFF, 10, No{});
    CTS(4);
    CTS(5);
    // ---- V^T = Wv . LN(tok)^T : swapped operands, lane (channel wn*160 + b*16 + rsel) holds tokens wm*32 + a*16 + 4g + r ----
    gemm(I10{}, I0{}, Yes{}, I0{}, acc, A_OFF, 10, Yes{});
    CTS(6);
    wait_vm<0>();   // the dead tail DMAs (zero-size descriptors still WRITE zeros) have landed: the ring is really free
    lds_barrier();  // every wave is done with the A tile and the ring: both become output staging
    // Outputs leave through LDS so that every global store instruction writes whole contiguous row segments (lane-owned
    // 80-B row pieces stored directly are 64 scattered 16-B writes per instruction: store-issue bound).
    //   A tile region : V^T [320 channels][64 tokens] (128-B rows, 16-B chunks XOR-swizzled by channel & 7)
    //   ring region   : two row-major [64 rows][640 B] tiles at a 656-B pitch (conflict-free 16-B writes)
    constexpr int RP = 656, ST0 = RING_OFF, ST1 = RING_OFF + 64 * RP;
    static_assert(ST1 + 64 * RP <= SCR_OFF, "staging tiles must fit in the ring region");
    // stage fragment rows: this lane's 8 columns cbase + 8q .. of fragment row a
#define STAGE_ROWS_ACC(base, v)                                                                                         \
  _Pragma("unroll") for (int a = 0; a < 2; a++) _Pragma("unroll") for (int q = 0; q < 5; q++) {                          \
    h8 o_;                                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 8; j++) o_[j] = (half_t)v[a][2 * q + (j >> 2)][j & 3];                         \
    *(h8*)(smem + (base) + (wm * 32 + a * 16 + rsel) * RP + (cbase + q * 8) * 2) = o_;                                   \
  }
    auto flush_rows = [&](int base, half_t* dst, int ld) {  // 2560 16-B chunks, 40 per row: a wave stores 1 KiB = 1.6 full rows
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const int t = tid + 256 * i, row = t / 40, c = t - row * 40;
        *(h8*)(dst + (long long)(m0 + row) * ld + c * 8) = *(const h8*)(smem + base + row * RP + c * 16);
      }
    };
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int q = 0; q < 5; q++) *(h8*)(smem + ST0 + (wm * 32 + a * 16 + rsel) * RP + (cbase + q * 8) * 2) = tokh[a][q];
    STAGE_ROWS_ACC(ST1, accq)
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 10; b++) {
        const int c = wn * 160 + b * 16 + rsel;
        const int tkn = wm * 32 + a * 16 + 4 * g;  // first of this lane's four tokens
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = (half_t)acc[a][b][r];
        *(h4*)(smem + A_OFF + c * 128 + (((tkn >> 3) ^ (c & 7)) << 4) + (tkn & 7) * 2) = o;
      }
    lds_barrier();
    flush_rows(ST0, p.tok_out, p.ld_tok);
    flush_rows(ST1, p.qk, p.ld_qk);
    {
      half_t* vb = p.vt + bsmp * p.s_vt + (m0 - bsmp * p.S);
#pragma unroll
      for (int i = 0; i < 10; i++) {  // 2560 16-B chunks: channel row t>>3, 8 tokens each: 128 B contiguous per row
        const int t = tid + 256 * i, c = t >> 3, part = t & 7;
        const h8 v = *(const h8*)(smem + A_OFF + c * 128 + ((part ^ (c & 7)) << 4));
        *(h8*)(vb + (long long)c * p.ld_vt + part * 8) = v;
      }
    }
    lds_barrier();  // staging tile 0 has been read by everyone
    STAGE_ROWS_ACC(ST0, acck)
#undef STAGE_ROWS_ACC
    lds_barrier();
    flush_rows(ST0, p.qk + C, p.ld_qk);
    CTS(7); CTS(8);
  } else {
  // =================================================================================================================
  f4 T[2][10];    // residual stream (fp32)
  f4 acc[2][10];  // stage accumulators
  f4 bv[10];      // per-column epilogue vector of the current stage (prefetched under the stage's GEMM)
  h8 raw[2][5];   // residual rows in flight (tok, later x)
  // ---- prologue: ao tile -> A tile ; residual 1 and the first bias in flight ; four weight tiles ahead ---------------
  {
    const rsrc_t ra = make_rsrc(p.ao);
#pragma unroll
    for (int i = 0; i < 10; i++) {  // 40 pieces of [8 rows][128 B]: piece j = k-tile j/8, rows (j%8)*8 ..
      const int j = wave + 4 * i, kt = j >> 3, row = (j & 7) * 8 + lrow;
      blds16(ra, (unsigned)((m0 + row) * p.ld_ao + cch * 8) * 2, kt * 128, smem + A_OFF + kt * A_KT + (j & 7) * 1024);
    }
  }
  load_rows_raw(p.tok, p.ld_tok, raw);
  load_cols(p.bso, bv);
  seg_begin(0);
  CTS(1);
  // ---- tok2 = ao . Wso^T + b + tok ------------------------------------------------------------------------------------
  gemm(I10{}, I0{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int b = 2 * q + (j >> 2), r = j & 3;
        T[a][b][r] = (float)raw[a][q][j] + (acc[a][b][r] + bv[b][r]);
      }
  CTS(2);
  layernorm_to_a(T);
  CTS(3);
  // ---- q = LN(tok2) . Wq^T  (scaled by softmax scale * log2 e) -----------------------------------------------------------
  gemm(I10{}, I0{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
  CTS(4);
  wait_vm<0>();   // the dead tail of segment 0 has landed: the ring region is free for the context keys / values
  lds_barrier();  // every wave is done reading LN(tok2) and the last weight tile
  // ---- cross attention over the sample's T context keys (helpers/attention.mojo:105-115) ------------------------------
  // Ring region: K tile 5 k-tiles x [80 key rows][128 B] (weight-operand layout) | V^T keys 0..63 [320 rows][128 B] |
  // V^T keys 64..79 [320 rows][32 B].  K row b*16 + ii holds key 32*(b>>1) + 8*(ii>>2) + 4*(b&1) + (ii&3) for b < 4
  // (output lane (g, r) of the key fragments 2kk, 2kk+1 then holds keys 32kk + 8g + 0..7: the A fragment of P.V k-step
  // kk) and key 64 + ii for b = 4 (k-step 2 pairs lane group g with keys 64 + 4g .. + 3).
  constexpr int KT_B = 80 * 128, V0_OFF = 5 * KT_B, V1_OFF = V0_OFF + 320 * 128;
  static_assert(V1_OFF + 320 * 32 <= NSLOT * SLOT, "context tiles must fit in the ring region");
  {
    const rsrc_t rk = make_rsrc(p.Kc + bsmp * p.sK), rv = make_rsrc(p.Vt + bsmp * p.sVt);
    const int nch = (p.T + 7) >> 3;
#pragma unroll
    for (int i = 0; i < 25; i++) {
      const int j = wave + 4 * i;  // 100 pieces of 1 KiB
      if (j < 50) {                // K: k-tile j/10, rows (j%10)*8 ..
        const int kt = j / 10, rho = (j - kt * 10) * 8 + lrow, b = rho >> 4, ii = rho & 15;
        const int kidx = b < 4 ? 32 * (b >> 1) + 8 * (ii >> 2) + 4 * (b & 1) + (ii & 3) : 64 + ii;
        blds16(rk, kidx < p.T ? (unsigned)(kidx * p.ldk + cch * 8) * 2 : PAD_OFF, kt * 128,
               smem + RING_OFF + kt * KT_B + (j - kt * 10) * 1024);
      } else if (j < 90) {         // V^T keys 0..63: rows (j-50)*8 ..
        const int row = (j - 50) * 8 + lrow;
        blds16(rv, cch < nch ? (unsigned)(row * p.ldvt + cch * 8) * 2 : PAD_OFF, 0, smem + RING_OFF + V0_OFF + (j - 50) * 1024);
      } else {                     // V^T keys 64..79: 32 rows x 32 B per piece
        const int row = (j - 90) * 32 + (lane >> 1), ch = 8 + (lane & 1);
        blds16(rv, ch < nch ? (unsigned)(row * p.ldvt + ch * 8) * 2 : PAD_OFF, 0, smem + RING_OFF + V1_OFF + (j - 90) * 1024);
      }
    }
  }
  store_a_tile(acc, p.qscale, 0.f, p.qscale, 0.f);  // q -> A tile while the context tiles fly
  load_cols(p.bco, bv);
  wait_vm<10>();  // the context tiles have landed (the 10 bias loads are younger)
  lds_barrier();  // q tile and context tiles visible
  // per wave: rows wm*32.., heads 4*wn .. 4*wn+3
#pragma unroll
  for (int hh = 0; hh < 4; hh++) {
    const int h = wn * 4 + hh, c0 = h * 5;  // first 16-B chunk of the head's 40 columns
    f4 sc[2][5];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) sc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 2; st++) {
      // k-step 0: chunks c0 .. c0+3 ; k-step 1: chunk c0+4 in lane group 0, zeros in the key operand elsewhere
      const int cg = st == 0 ? c0 + g : c0 + 4;
      const int ktile = cg >> 3, cpos = cg & 7;
      h8 qa[2], kb[5];
#pragma unroll
      for (int a = 0; a < 2; a++) qa[a] = *(const h8*)(smem + A_OFF + ktile * A_KT + a_rd + a * 2048 + ((cpos ^ key) << 4));
#pragma unroll
      for (int b = 0; b < 5; b++) {
        kb[b] = *(const h8*)(smem + RING_OFF + ktile * KT_B + (b * 16 + rsel) * 128 + ((cpos ^ key) << 4));
        if (st == 1 && g != 0) kb[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) sc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb[b], qa[a], sc[a][b], 0, 0, 0);
    }
    // softmax over the keys of each query row: 20 in-lane scores x 4 lane groups (max subtraction, App.A D6)
    h8 pf[2][3];
    float rinv[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float mx = -1.0e30f;
#pragma unroll
      for (int b = 0; b < 5; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kidx = b < 4 ? 32 * (b >> 1) + 8 * g + 4 * (b & 1) + r : 64 + 4 * g + r;
          if (kidx >= p.T) sc[a][b][r] = -1.0e30f;
          mx = fmaxf(mx, sc[a][b][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16));
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          half_t ph = (half_t)0.f;
          if (kk < 2 || j < 4) ph = (half_t)__builtin_amdgcn_exp2f(sc[a][kk < 2 ? 2 * kk + (j >> 2) : 4][j & 3] - mx);
          o[j] = ph;
          sum += (float)ph;  // the normaliser sums the SAME rounded probabilities the MFMA multiplies
        }
        pf[a][kk] = o;
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      rinv[a] = 1.f / sum;
    }
    // O_h = P . V_h : channel rows h*40 + b*16 + rsel of the V^T tiles (rows past the head's 40 give discarded outputs)
    f4 oc[2][3];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) oc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      h8 vb[3];
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const int row = min(h * 40 + b * 16 + rsel, C - 1);
        if (kk < 2) vb[b] = *(const h8*)(smem + RING_OFF + V0_OFF + row * 128 + (((kk * 4 + g) ^ (row & 7)) << 4));
        else {
          const h4 lo = *(const h4*)(smem + RING_OFF + V1_OFF + row * 32 + g * 8);
          vb[b] = h8{lo[0], lo[1], lo[2], lo[3], 0, 0, 0, 0};
        }
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) oc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[b], pf[a][kk], oc[a][b], 0, 0, 0);
    }
    // The head's output overwrites the head's q columns in the A tile.  Columns h*40 .. h*40+39 of rows wm*32 .. +31 are
    // read (as q) by THIS wave only - heads 4wn .. 4wn+3 belong to the column wave, the rows to the row wave - and its
    // QK^T for this head is done, so no barrier is needed.  lane (g, r) of fragment b holds channel h*40 + b*16 + 4g + r.
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int row = wm * 32 + a * 16 + rsel;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const int d = b * 16 + 4 * g;
        if (d < 40) {
          const int col = h * 40 + d;
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (half_t)(oc[a][b][r] * rinv[a]);
          *(h4*)(smem + A_OFF + (col >> 6) * A_KT + row * 128 + ((((col >> 3) & 7) ^ key) << 4) + (col & 7) * 2) = o;
        }
      }
    }
  }
  lds_barrier();  // attention output complete in the A tile ; the ring region is free again
  CTS(5);
  seg_begin(1);
  // ---- tok3 = attn . Wco^T + b + tok2 -------------------------------------------------------------------------------
  gemm(I10{}, I1{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) T[a][b] += acc[a][b] + bv[b];
  layernorm_to_a(T);
  CTS(6);
  // ---- GEGLU feed-forward: ten chunks of 128 hidden units; the second GEMM accumulates over the chunks ------------------
  f4 acc2[2][10];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) acc2[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#if TSD_CHAIN_PIPE
  {
    // ONE software pipeline over the 140 GEGLU tiles (round 3).  Round 2 ran per chunk GEMM-1 (10 tiles) -> a * gelu(g) ->
    // GEMM-2 (4 tiles) as three serial phases: every GEMM call opened with a read-only step and closed with an MFMA-only step,
    // and the 32 activations per lane ran with the matrix pipe idle - 9.8 K ticks per chunk for 3.8 K of MFMA.
    // Here a step = barrier of tile gt, the MFMAs of tile gt-1 (whatever GEMM it belongs to) from one fragment register set,
    // the fragment reads of tile gt into the other, the DMA of tile gt+4 - straight through the walk
    //   G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(9) G2(8) | G2(9)
    // and chunk j's activation is computed 8 values per step under the 20 MFMAs of G2(j-1)'s tiles (the accumulator of G1(j)
    // is complete by then and G1(j+1) has not started), kept in 16 registers, and written to the activation tile at the second
    // step of G1(j+1) - two barriers after G2(j-1)'s last read of that tile, nine steps before G2(j)'s first.  Same products
    // in the same order as round 2: bit-identical results.
    f4 b1v[8];
    h8 oreg[2][2];
    h8 afS[2][2], wfS[2][10];
    auto load_b1 = [&](int jc) {
      const float* bp = p.b1 + jc * 256 + wn * 128 + g * 32;
#pragma unroll
      for (int b = 0; b < 8; b++) b1v[b] = *(const f4*)(bp + b * 4);
    };
    auto act_unit = [&](int u, int j0, int j1) {  // elements j0..j1-1 of unit u = (a, q): 8 activations = one 16-B chunk
      const int a = u >> 1, q = u & 1;
#pragma unroll
      for (int j = j0; j < j1; j++) {
        const int b = q * 4 + (j >> 1), r = (j & 1) * 2;
        oreg[a][q][j] = (half_t)((acc[a][b][r] + b1v[b][r]) * gelu_tanh_c(acc[a][b][r + 1] + b1v[b][r + 1]));
      }
    };
    auto write_act = [&]() {
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int q = 0; q < 2; q++)
          *(h8*)(smem + ACT_OFF + wn * A_KT + (wm * 32 + a * 16 + rsel) * 128 + (((g * 2 + q) ^ key) << 4)) = oreg[a][q];
    };
    constexpr int F_ZERO = 1, F_EXTRA = 2, F_WRITE = 4, F_PLAIN = 8;
    // PK / CK: kind of the previous / current tile (0 none, 1 = GEMM-1 tile: 256 rows, FN 8, A from the LN tile ; 2 = GEMM-2
    // tile: 320 rows, FN 10, A from the activation tile) ; KC: index of the current tile in its GEMM ; SET: fragment register
    // set the current tile's fragments go to ; FILL: activation unit computed under this step's MFMAs (-1 none)
    auto gstep = [&](auto pk_c, auto ck_c, auto kc_c, auto set_c, auto fill_c, auto flags_c) {
      constexpr int PK = decltype(pk_c)::value, CK = decltype(ck_c)::value, KC = decltype(kc_c)::value, SET = decltype(set_c)::value;
      constexpr int FILL = decltype(fill_c)::value, FLAGS = decltype(flags_c)::value;
      constexpr int FNp = PK == 1 ? 8 : 10, FNc = CK == 1 ? 8 : 10;
      bool g1n = false, liven = false;
      int off = 0, s4 = 0;
      const char *sA = smem, *sW = smem;
      if constexpr (CK != 0) {
        int younger = 0;
#pragma unroll
        for (int d = 1; d <= 3; d++) { bool yg1, ylive; tile_off(1, gt + d, yg1, ylive); younger += yg1 ? 4 : 5; }
        constexpr int EX = (FLAGS & F_EXTRA) ? 8 : 0;
        switch (younger) {
          case 12: wait_vm<12 + EX>(); break;
          case 13: wait_vm<13 + EX>(); break;
          case 14: wait_vm<14 + EX>(); break;
          default: wait_vm<15 + EX>(); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads of tile gt-1 (and any activation-tile writes) are complete
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sA = smem + (CK == 1 ? A_OFF : ACT_OFF) + (KC >> 1) * A_KT + a_rd + ((((KC & 1) * 4 + g) ^ key) << 4);
        sW = smem + RING_OFF + sl * SLOT + wn * (FNc == 10 ? 10240 : 8192) + w_rd;
        off = tile_off(1, gt + 4, g1n, liven);
        s4 = sl == 0 ? 4 : sl - 1;
      }
      if constexpr ((FLAGS & F_WRITE) != 0) write_act();
      h8 (&af)[2] = afS[SET];
      h8 (&wf)[10] = wfS[SET];
      const h8 (&paf)[2] = afS[SET ^ 1];
      const h8 (&pwf)[10] = wfS[SET ^ 1];
      if constexpr (PK == 0) {
        if constexpr (CK != 0) {
#pragma unroll
          for (int a = 0; a < 2; a++) af[a] = *(const h8*)(sA + a * 2048);
#pragma unroll
          for (int b = 0; b < FNc; b++) wf[b] = *(const h8*)(sW + b * 1024);
#pragma unroll
          for (int i = 0; i < 5; i++) if (!(g1n && i == 4)) piece(off, g1n, liven, s4, i);
        }
      } else {
        constexpr int NM = 2 * FNp, NR = CK != 0 ? 2 + FNc : 0;
#pragma unroll
        for (int q = 0; q < NM; q++) {
          const int b = q >> 1, a = q & 1;
          const f4 c0 = (FLAGS & F_ZERO) ? f4{0.f, 0.f, 0.f, 0.f} : (PK == 1 ? acc[a][b] : acc2[a][b]);
          const f4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[b], paf[a], c0, 0, 0, 0);
          if (PK == 1) acc[a][b] = d; else acc2[a][b] = d;
          __builtin_amdgcn_sched_barrier(0);
          if (q < NR) {
            if (q < 2) af[q] = *(const h8*)(sA + q * 2048);
            else wf[q - 2] = *(const h8*)(sW + (q - 2) * 1024);
          }
          if (CK != 0 && (q + 1) % (NM / 5) == 0 && (q + 1) / (NM / 5) - 1 < 5) {
            const int i = (q + 1) / (NM / 5) - 1;
            if (!(g1n && i == 4)) piece(off, g1n, liven, s4, i);
          }
          // one activation after every second MFMA (8 per step)
          if (FILL >= 0 && (q & 1) && (q >> 1) < 8) act_unit(FILL, q >> 1, (q >> 1) + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr ((FLAGS & F_PLAIN) != 0) {  // chunk 0: nothing to hide its activation under
#pragma unroll
        for (int u = 0; u < 4; u++) act_unit(u, 0, 8);
        write_act();
      }
      if constexpr (CK != 0) {
        gt++;
        sl = sl == 4 ? 0 : sl + 1;
      }
    };
    using IM1 = std::integral_constant<int, -1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using I4 = std::integral_constant<int, 4>;
    using I5 = std::integral_constant<int, 5>;
    using I6 = std::integral_constant<int, 6>;
    using I7 = std::integral_constant<int, 7>;
    using I9 = std::integral_constant<int, 9>;
    // ---- G1(0): tiles 0..9 (bias of chunk 0 in flight under its first four steps) ----
    load_b1(0);
    gstep(I0{}, I1{}, I0{}, I0{}, IM1{}, I2{});                 // flags: EXTRA
    gstep(I1{}, I1{}, I1{}, I1{}, IM1{}, I3{});                 // ZERO (first product of the chunk) + EXTRA
    gstep(I1{}, I1{}, I2{}, I0{}, IM1{}, I2{});
    gstep(I1{}, I1{}, I3{}, I1{}, IM1{}, I2{});
    gstep(I1{}, I1{}, I4{}, I0{}, IM1{}, I0{});
    gstep(I1{}, I1{}, I5{}, I1{}, IM1{}, I0{});
    gstep(I1{}, I1{}, I6{}, I0{}, IM1{}, I0{});
    gstep(I1{}, I1{}, I7{}, I1{}, IM1{}, I0{});
    gstep(I1{}, I1{}, I8{}, I0{}, IM1{}, I0{});
    gstep(I1{}, I1{}, I9{}, I1{}, IM1{}, I0{});
    for (int j = 1; j < 10; j++) {
      // ---- G1(j): the first step finishes the previous tile (G1(0)'s last for j = 1: then chunk 0's activation runs in the
      // open; else G2(j-2)'s last, hiding unit 3 of chunk j-1) ; the second publishes chunk j-1's activations ----
      if (j == 1) gstep(I1{}, I1{}, I0{}, I0{}, IM1{}, I8{});   // PLAIN: all of chunk 0's activation + write
      else gstep(I2{}, I1{}, I0{}, I0{}, I3{}, I0{});
      if (j == 1) gstep(I1{}, I1{}, I1{}, I1{}, IM1{}, I1{});   // ZERO
      else gstep(I1{}, I1{}, I1{}, I1{}, IM1{}, I5{});          // ZERO + WRITE
      gstep(I1{}, I1{}, I2{}, I0{}, IM1{}, I0{});
      gstep(I1{}, I1{}, I3{}, I1{}, IM1{}, I0{});
      gstep(I1{}, I1{}, I4{}, I0{}, IM1{}, I0{});
      load_b1(j);                                               // used from the second step of G2(j-1) on: six steps from here
      gstep(I1{}, I1{}, I5{}, I1{}, IM1{}, I2{});               // EXTRA for four steps
      gstep(I1{}, I1{}, I6{}, I0{}, IM1{}, I2{});
      gstep(I1{}, I1{}, I7{}, I1{}, IM1{}, I2{});
      gstep(I1{}, I1{}, I8{}, I0{}, IM1{}, I2{});
      gstep(I1{}, I1{}, I9{}, I1{}, IM1{}, I0{});
      // ---- G2(j-1): chunk j's activation units 0..2 under its tiles' MFMAs ----
      gstep(I1{}, I2{}, I0{}, I0{}, IM1{}, I0{});
      gstep(I2{}, I2{}, I1{}, I1{}, I0{}, I0{});
      gstep(I2{}, I2{}, I2{}, I0{}, I1{}, I0{});
      gstep(I2{}, I2{}, I3{}, I1{}, I2{}, I0{});
    }
    // ---- G2(8)'s last tile with unit 3 of chunk 9, publish, then G2(9) ----
    gstep(I2{}, I0{}, I0{}, I0{}, I3{}, I0{});
    // that step had no tile barrier (CK = 0): G2(8)'s last reads of the activation tile by the OTHER column wave must have retired
    // before this wave overwrites its rows with chunk 9 (inside the loop the publish sits two barriers behind the last read)
    lds_barrier();
    write_act();
    gstep(I0{}, I2{}, I0{}, I0{}, IM1{}, I0{});
    gstep(I2{}, I2{}, I1{}, I1{}, IM1{}, I0{});
    gstep(I2{}, I2{}, I2{}, I0{}, IM1{}, I0{});
    gstep(I2{}, I2{}, I3{}, I1{}, IM1{}, I0{});
    gstep(I2{}, I0{}, I0{}, I0{}, IM1{}, I0{});
  }
#else
  for (int jc = 0; jc < 10; jc++) {
    if (jc == 5) CTS(10);
    f4 b1v[8];  // this chunk's (a, g) bias pairs: in flight under the chunk's first GEMM
    {
      const float* bp = p.b1 + jc * 256 + wn * 128 + g * 32;
#pragma unroll
      for (int b = 0; b < 8; b++) b1v[b] = *(const f4*)(bp + b * 4);
    }
    gemm(I8{}, I1{}, Yes{}, I8{}, acc, A_OFF, 10, No{});  // (a, g) interleaved: 256 columns
    if (jc == 5) CTS(11);
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int row = wm * 32 + a * 16 + rsel;
#pragma unroll
      for (int q = 0; q < 2; q++) {  // 16 activations per lane and fragment row = two 16-B chunks
        h8 o;
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int b = q * 4 + (j >> 1), r = (j & 1) * 2;
          o[j] = (half_t)((acc[a][b][r] + b1v[b][r]) * gelu_tanh_c(acc[a][b][r + 1] + b1v[b][r + 1]));
        }
        *(h8*)(smem + ACT_OFF + wn * A_KT + row * 128 + (((g * 2 + q) ^ key) << 4)) = o;
      }
    }
    // the tile barrier that opens the next GEMM makes the activations visible (and the 10 tile barriers of the next
    // chunk's first GEMM separate this chunk's reads from the next chunk's writes)
    if (jc == 5) CTS(12);
    gemm(I10{}, I1{}, No{}, I0{}, acc2, ACT_OFF, 4, No{});
    if (jc == 5) CTS(13);
  }
#endif
  CTS(7);
  load_cols(p.b2, bv);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) T[a][b] += acc2[a][b] + bv[b];
  // tok4 -> A tile (every wave passed the last GEGLU-1 tile long ago: the A tile is free)
  store_a_tile(T, 1.f, 0.f, 1.f, 0.f);
  // ---- out = tok4 . Wout^T + b + x --------------------------------------------------------------------------------
  load_cols(p.bout, bv);
  load_rows_raw(p.x, p.ld_x, raw);
  gemm(I10{}, I1{}, Yes{}, I20{}, acc, A_OFF, 10, No{});
  CTS(8);
  float gs1[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's 4 groups of 10 channels
  lds_barrier();  // every wave is done with the A tile: it (and the idle activation tile behind it) stages the output rows
  constexpr int RP = 656;  // row pitch of the row-major staging tile: conflict-free 16-B writes, whole-row global stores
  static_assert(A_OFF + 64 * RP <= RING_OFF, "output staging must stay clear of the ring (dead DMAs may still land there)");
#pragma unroll
  for (int a = 0; a < 2; a++) {
#pragma unroll
    for (int q = 0; q < 5; q++) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int b = 2 * q + (j >> 2), r = j & 3;
        o[j] = (half_t)((float)raw[a][q][j] + (acc[a][b][r] + bv[b][r]));
        const float f = (float)o[j];
        const int grp = (q * 8 + j) / 10;
        gs1[grp] += f; gs2[grp] += f * f;
      }
      *(h8*)(smem + A_OFF + (wm * 32 + a * 16 + rsel) * RP + (cbase + q * 8) * 2) = o;
    }
  }
  lds_barrier();
#pragma unroll
  for (int i = 0; i < 10; i++) {  // 2560 16-B chunks, 40 per row: a wave instruction stores 1 KiB of consecutive row bytes
    const int t = tid + 256 * i, row = t / 40, c = t - row * 40;
    *(h8*)(p.out + (long long)(m0 + row) * p.ld_out + c * 8) = *(const h8*)(smem + A_OFF + row * RP + c * 16);
  }
  if (p.gn_part) {
    // GroupNorm(32) statistics of the rounded output for the consumer (one 32-row slab per wave row block)
#pragma unroll
    for (int k = 0; k < 4; k++) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { gs1[k] += __shfl_xor(gs1[k], o); gs2[k] += __shfl_xor(gs2[k], o); }
    }
    if (rsel == 0) {
      const int mrow = m0 + wm * 32, bb = mrow / p.S, slab = (mrow - bb * p.S) >> 5;
      float* ob = p.gn_part + (((long long)bb * p.gn_nslab + slab) * 32 + wn * 16 + g * 4) * 2;
#pragma unroll
      for (int k = 0; k < 4; k++) *(f2*)(ob + k * 2) = f2{gs1[k], gs2[k]};
    }
  }
  }
  wait_vm<0>();  // the dead tail DMAs have landed before the workgroup's LDS is released
  CTS(9);
  CTS(15);
}

#ifdef TSD_CHAIN_TS
extern "C" int tsd_debug_chain_ts(unsigned long long* out, int n, int kind) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_chain_ts), (size_t)n * 8, (size_t)kind * 1024 * 16 * 8) == hipSuccess ? 0 : -1;
}
#endif

// ---- host side ----------------------------------------------------------------------------------------------------
size_t attn_tail_stream_bytes() { return (size_t)STREAM_BYTES; }

// debug / A-B switch of ONE context: 1 = fused head / tail kernels at the 64x64 level (default), 0 = the op-by-op graph; returns the old value
extern "C" int tsd_debug_set_fused_attention(tsd_ctx* ctx, int on) {
  if (!ctx) return TSD_E_ARG;
  const int old = ctx->opt.chain;
  ctx->opt.chain = on ? 1 : 0;
  ctx->opt.gen++;
  return old;
}
bool attn_tail_supported(const tsd_ctx* ctx, int C_, int d, int heads, int T, int64_t M, int S) {
  return ctx->opt.chain && C_ == 320 && d == 40 && heads == 8 && T >= 1 && T <= 80 && S % 64 == 0 && M % 64 == 0 && M < (1 << 24);
}

// pack the six weight matrices of one attention block (fp16, reference-packed [N][K] with the GEGLU rows interleaved)
int launch_attn_tail_pack(tsd_ctx* ctx, const half_t* Wso, int ld_so, const half_t* Wq, int ld_q, const half_t* Wco, int ld_co,
                          const half_t* W1, int ld_1, const half_t* W2, int ld_2, const half_t* Wout, int ld_out, half_t* dst) {
  if (ld_so < 320 || ld_q < 320 || ld_co < 320 || ld_1 < 320 || ld_2 < 1280 || ld_out < 320)
    TSD_FAIL(TSD_E_SHAPE, "attention tail: unexpected weight pitches");
  if (!ctx->launch()) return TSD_OK;
  const int n = STREAM_BYTES / 16;
  hipLaunchKernelGGL(k_attn_tail_pack, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, PackSrc{Wso, ld_so}, PackSrc{Wq, ld_q},
                     PackSrc{Wco, ld_co}, PackSrc{W1, ld_1}, PackSrc{W2, ld_2}, PackSrc{Wout, ld_out}, dst);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_attn_tail(tsd_ctx* ctx, const AttnTailArgs& a) {
  if (!attn_tail_supported(ctx, a.C, a.d, a.heads, a.T, a.M, a.S)) TSD_FAIL(TSD_E_SHAPE, "attention tail: unsupported shape");
  if (!a.wstream) TSD_FAIL(TSD_E_ARG, "attention tail: weights were not packed");
  if (a.ld_ao % 8 || a.ld_tok % 8 || a.ld_x % 8 || a.ld_out % 8 || a.ldk % 8 || a.ldvt % 8 || a.ld_ao < 320 || a.ld_tok < 320 ||
      a.ld_x < 320 || a.ld_out < 320 || a.ldk < 320 || a.ldvt < ((a.T + 7) & ~7))
    TSD_FAIL(TSD_E_SHAPE, "attention tail: misaligned or too narrow pitches");
  // the ao tile is DMA'd through ONE descriptor based at a.ao with 32-bit byte offsets (m0 + row) * ld_ao * 2
  if ((long long)a.M * a.ld_ao * 2 >= 0x7ffffff0LL) TSD_FAIL(TSD_E_SHAPE, "attention tail: ao operand exceeds the 2 GiB addressing window");
  if (!a.ao || !a.tok || !a.x || !a.out || !a.Kc || !a.Vt || !a.bso || !a.bco || !a.b1 || !a.b2 || !a.bout)
    TSD_FAIL(TSD_E_ARG, "attention tail: NULL operand");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_CHAIN, (int)a.M, a.C, 0, 1);
  TailK k;
  k.ao = a.ao; k.tok = a.tok; k.x = a.x; k.out = a.out;
  k.ld_ao = a.ld_ao; k.ld_tok = a.ld_tok; k.ld_x = a.ld_x; k.ld_out = a.ld_out;
  k.wstream = a.wstream;
  k.bso = a.bso; k.bco = a.bco; k.b1 = a.b1; k.b2 = a.b2; k.bout = a.bout;
  k.Kc = a.Kc; k.ldk = a.ldk; k.sK = a.sK; k.Vt = a.Vt; k.ldvt = a.ldvt; k.sVt = a.sVt;
  k.T = a.T; k.M = (int)a.M; k.S = a.S;
  k.qscale = a.scale * 1.4426950408889634f; k.eps = a.eps;
  k.gn_part = a.gn_part; k.gn_nslab = a.gn_nslab;
  k.gn_stats = nullptr; k.b_in = nullptr; k.tok_out = nullptr; k.qk = nullptr; k.ld_qk = 0; k.vt = nullptr; k.ld_vt = 0; k.s_vt = 0;
  static std::atomic<unsigned long long> attr{0};  // one bit per device (setting the attribute twice is harmless; the mask is only a shortcut)
  if (!((attr.load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)attn_chain_kernel<KIND_TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(attn_chain_kernel<KIND_TAIL>, dim3((unsigned)(a.M / BM)), dim3(256), LDS_BYTES, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- head of the block: GroupNorm-apply -> conv_in -> tok ; LayerNorm -> q, k, V^T ------------------------------------
size_t attn_head_stream_bytes() { return (size_t)HEAD_STREAM_BYTES; }

int launch_attn_head_pack(tsd_ctx* ctx, const half_t* Wc, int ld_c, const half_t* Win, int ld_in, half_t* dst) {
  if (ld_c < 320 || ld_in < 320) TSD_FAIL(TSD_E_SHAPE, "attention head: unexpected weight pitches");
  if (!ctx->launch()) return TSD_OK;
  const int n = HEAD_STREAM_BYTES / 16;
  hipLaunchKernelGGL(k_attn_head_pack, dim3(ceil_div(n, 256)), dim3(256), 0, ctx->stream, PackSrc{Wc, ld_c}, PackSrc{Win, ld_in}, dst);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_attn_head(tsd_ctx* ctx, const AttnHeadArgs& a) {
  if (!attn_tail_supported(ctx, 320, 40, 8, 1, a.M, a.S)) TSD_FAIL(TSD_E_SHAPE, "attention head: unsupported shape");
  if (!a.wstream || !a.gn_stats) TSD_FAIL(TSD_E_ARG, "attention head: weights were not packed / statistics missing");
  if (a.ld_x % 8 || a.ld_tok % 8 || a.ld_qk % 8 || a.ld_vt % 8 || a.ld_x < 320 || a.ld_tok < 320 || a.ld_qk < 640 || a.ld_vt < a.S)
    TSD_FAIL(TSD_E_SHAPE, "attention head: misaligned or too narrow pitches");
  if (!a.x || !a.tok || !a.qk || !a.vt || !a.b_in) TSD_FAIL(TSD_E_ARG, "attention head: NULL operand");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_CHAIN, (int)a.M, 320, 1, 1);
  TailK k = {};
  k.x = a.x; k.ld_x = a.ld_x; k.tok_out = a.tok; k.ld_tok = a.ld_tok; k.qk = a.qk; k.ld_qk = a.ld_qk;
  k.vt = a.vt; k.ld_vt = a.ld_vt; k.s_vt = a.s_vt;
  k.wstream = a.wstream; k.b_in = a.b_in; k.gn_stats = a.gn_stats;
  k.M = (int)a.M; k.S = a.S; k.eps = a.eps; k.T = 1;
  static std::atomic<unsigned long long> attr{0};  // one bit per device
  if (!((attr.load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)attn_chain_kernel<KIND_HEAD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(attn_chain_kernel<KIND_HEAD>, dim3((unsigned)(a.M / BM)), dim3(256), LDS_BYTES, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
