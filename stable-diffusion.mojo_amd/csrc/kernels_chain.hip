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
//   * LayerNorm statistics are an in-lane sum + two row swaps (v_permlane16/32_swap) + one LDS exchange between the four
//     column waves (per-wave mean / M2 merged exactly);
//   * the output's GroupNorm statistics (for the next residual block) come from the rounded values in registers.
// MFMA: v_mfma_f32_16x16x32_f16 with swapped operands (D = Wfrag x Afrag^T).  Round 4: 4 waves as 1(M) x 4(N) - a wave tile is
// all 64 rows x 80 columns (FM = 4, FN = 5), one wave per SIMD - and every wave streams ITS OWN quarter of the weight tiles, so
// the tile loops run without a barrier per tile (DESIGN.md 4.2 "Round 4").  Two things in this file exist because of what the
// ISA listings showed: pin_here() (hipcc moves side-effect-free arithmetic to its use, whatever sched_barriers it was written
// between) and the static tile walk of the feed-forward pipeline.
#include <math.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

// Ablation builds of the GEGLU pipeline (timing only, results wrong): bit 0 no tile barrier, bit 1 no fragment reads, bit 2 no MFMAs,
// bit 3 no DMA pieces, bit 4 no activation arithmetic, bit 5 no counted DMA wait
#ifndef TSD_CHAIN_ABL
#define TSD_CHAIN_ABL 0
#endif
#ifndef TSD_CHAIN_TILE_BARRIER
#define TSD_CHAIN_TILE_BARRIER 0  // 1 = a barrier at every tile step (diagnostic: the round-3 cadence)
#endif
#ifndef TSD_CHAIN_ARES
#define TSD_CHAIN_ARES 0  // GEMM-1 k-tiles whose A fragments stay in registers through the feed-forward (0 = none)
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
constexpr int RING_OFF = ACT_OFF + 16384, SLOT = 20480, NSLOT = 5;  // ring of [4 quarters][80 rows][64 B] weight tiles
constexpr int SCR_OFF = RING_OFF + NSLOT * SLOT;      // LayerNorm exchange: [64 rows][4 column waves] x (mean, M2)
constexpr int LDS_BYTES = SCR_OFF + 2048;
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
  // x * sigmoid(2u), u = sqrt(2/pi) (x + 0.044715 x^3): the exponent -2u log2(e) = x * (k1 x^2 + k0) as mul, fma, mul (hipcc kept four
  // operations for c2 * (x + 0.044715 * x * x * x)); these run next to the MFMAs of the feed-forward pipeline, where every VALU
  // instruction adds its issue time (profiles/r04_chain_geglu_ablation.txt)
  const float k0 = -2.f * 0.7978845608028654f * 1.4426950408889634f, k1 = k0 * 0.044715f;
  const float x2 = x * x;
  const float t = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x2, k1, k0));
  return x * __builtin_amdgcn_rcpf(1.f + t);
}
// Counted waits and the callers' "EXTRA" loads (round 4, late - a real bug): a stage whose bias / residual loads are issued BEHIND the
// first weight tiles used to add their number to the first four counted waits (vmcnt(15 + 25): "tile gt has landed, the three
// younger tiles and my 25 loads may stay in flight").  The 25 was the SOURCE's count; hipcc merges the twenty 8-byte residual loads
// pairwise where their addresses are adjacent and issues 17 instructions, so the wait let 8 OLDER instructions - tile gt's own
// pieces - stay in flight too.  The tile had practically always landed anyway: 14 of 1500 fifty-step runs differed on one box, none
// with the count removed (same speed: the younger tiles and loads have long been issued).  STRICT_WAIT = 1 (default): the counted
// waits never include loads the compiler is free to merge, split or move - they are simply waited for as well.  0 = the old
// counting (kept to reproduce the failure), 2 = every counted wait is vmcnt(0) (hazard hunting).
#ifndef TSD_CHAIN_STRICT_WAIT
#define TSD_CHAIN_STRICT_WAIT 1
#endif
// DMA pieces of younger tiles (4 per GEMM-1 quarter tile, 5 per full one) + OTHER plain loads may stay in flight; the accounting
// travels into the assembly and tools/isa_lint.py checks it against what hipcc emitted (lds_dma.h)
template <int DMA, int OTHER = 0>
__device__ __forceinline__ void wait_vm() {
  if constexpr (TSD_CHAIN_STRICT_WAIT >= 2) wait_vm_counted<0>(); else wait_vm_counted<DMA, OTHER, 5, 4>();
}
// An opaque use-and-redefine of a value: whatever computes it must be issued before this point and whatever consumes it after - it
// ties side-effect-free arithmetic to its place between the MFMAs.  (A __device__ function so that the host pass never sees the "v"
// constraint.)
__device__ __forceinline__ unsigned pin_here(unsigned w) { asm volatile("" : "+v"(w)); return w; }
// Reductions over the four 16-lane rows of a wave (lanes with the same lane & 15) without LDS round trips: gfx950's
// v_permlane16_swap / v_permlane32_swap exchange rows of two registers; with both holding the same value, a' + b' is the pairwise
// row sum in EVERY lane.  __shfl_xor lowers to ds_bpermute_b32 (an LDS round trip per dependent step: the LayerNorm and softmax
// statistics are chains of four).  (The __builtin_amdgcn_permlane*_swap builtins, given the same value for both operands, came out
// as a' + a' with hipcc 7.2 - hence the asm; s_nop 1 covers the VALU-write -> permlane hazard the assembler does not see.)
__device__ __forceinline__ void rows_swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void rows_swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ float rows_sum(float v) {  // sum over the wave's four rows, in every lane
  float a = v, b = v;
  rows_swap16(a, b);
  a += b; b = a;
  rows_swap32(a, b);
  return a + b;
}
__device__ __forceinline__ float rows_max(float v) {
  float a = v, b = v;
  rows_swap16(a, b);
  a = fmaxf(a, b); b = a;
  rows_swap32(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ void lds_barrier() {
  tsd_jitter();
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
//   image byte  quarter*QB + rho*64 + pc*16  <-  W[n(quarter, rho)][kofs + 32t + 8*(pc ^ wswz(rho)) .. +8]
// with n(quarter, rho) = quarter*NQ + (ii>>2)*(NQ/4) + fn*4 + (ii&3), fn = rho>>4, ii = rho&15 (NQ = rows per quarter = the
// columns of one wave: 80, or 64 for a GEGLU-1 tile): output lane (g, r) of fragment b then owns column
// quarter*NQ + g*(NQ/4) + b*4 + r - NQ/4 consecutive columns per lane.
struct PackSrc { const half_t* w; int ldw; };
__global__ void k_attn_tail_pack(PackSrc so, PackSrc q, PackSrc co, PackSrc w1, PackSrc w2, PackSrc wo, half_t* dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;  // 16-B chunk index in the stream
  if (ci >= STREAM_BYTES / 16) return;
  int byte = ci * 16;
  const half_t* W; int ldw, row0 = 0, kofs = 0, NH, t;
  if (byte < SEG0_BYTES) {
    const int ti = byte / TILE_FULL; byte -= ti * TILE_FULL;
    W = ti < 10 ? so.w : q.w; ldw = ti < 10 ? so.ldw : q.ldw; t = ti % 10; NH = 80;
  } else {
    byte -= SEG0_BYTES;
    if (byte < 10 * TILE_FULL) { t = byte / TILE_FULL; byte -= t * TILE_FULL; W = co.w; ldw = co.ldw; NH = 80; }
    else if (byte < 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES) {
      byte -= 10 * TILE_FULL;
      const int jc = byte / FFN_CHUNK_BYTES; byte -= jc * FFN_CHUNK_BYTES;
      if (byte < 10 * TILE_G1) { t = byte / TILE_G1; byte -= t * TILE_G1; W = w1.w; ldw = w1.ldw; row0 = jc * 256; NH = 64; }
      else { byte -= 10 * TILE_G1; t = byte / TILE_FULL; byte -= t * TILE_FULL; W = w2.w; ldw = w2.ldw; kofs = jc * 128; NH = 80; }
    } else {
      byte -= 10 * TILE_FULL + 10 * FFN_CHUNK_BYTES;
      t = byte / TILE_FULL; byte -= t * TILE_FULL; W = wo.w; ldw = wo.ldw; NH = 80;
    }
  }
  const int half = byte / (NH * 64), rb = byte - half * NH * 64;  // "half" = quarter (0..3), NH = its rows
  const int rho = rb >> 6, pc = (rb >> 4) & 3;
  const int fn = rho >> 4, ii = rho & 15;
  const int n = row0 + half * NH + (ii >> 2) * (NH / 4) + fn * 4 + (ii & 3);
  const int k = kofs + 32 * t + 8 * (pc ^ wswz(rho));
  *(h8*)(dst + (size_t)ci * 8) = *(const h8*)(W + (size_t)n * ldw + k);
}

// head stream: conv_in, then the q / k / v row blocks of in_proj (rows 0..319 / 320..639 / 640..959), all K = 320.
// The v tiles keep the IDENTITY row order (LDS row rho of quarter h = weight row h*80 + rho): that stage runs with swapped
// MFMA operands and wants consecutive channels across a fragment's 16 lanes.
__global__ void k_attn_head_pack(PackSrc cin, PackSrc inproj, half_t* dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= HEAD_STREAM_BYTES / 16) return;
  int byte = ci * 16;
  const int ti = byte / TILE_FULL; byte -= ti * TILE_FULL;
  const int mat = ti / 10, t = ti - mat * 10;
  const half_t* W = mat == 0 ? cin.w : inproj.w;
  const int ldw = mat == 0 ? cin.ldw : inproj.ldw, row0 = mat == 0 ? 0 : (mat - 1) * 320;
  const int half = byte / 5120, rb = byte - half * 5120;  // quarter 0..3: 80 rows of 64 B
  const int rho = rb >> 6, pc = (rb >> 4) & 3, fn = rho >> 4, ii = rho & 15;
  const int n = row0 + half * 80 + (mat == 3 ? rho : (ii >> 2) * 20 + fn * 4 + (ii & 3));
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
  const int wn = wave;  // 1(M) x 4(N) wave grid: every wave owns all 64 rows and 80 of the 320 columns (FM = 4, FN = 5)
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
  // piece i (of 5 per wave; 4 for a GEMM-1 tile) of tile gi -> ring slot.  Round 4: a wave fetches ITS OWN quarter of the tile image
  // (bytes [wn * Q, wn * Q + Q), Q = 5 KiB / 4 KiB - exactly the weight rows its MFMAs read), so a wave's counted vmcnt wait is all it
  // needs before it reads a tile, and a ring slot is only ever rewritten by the wave that read it: the tile loops run WITHOUT a
  // barrier per tile (one barrier where the shared A operand changes hands instead of one per step: 140 -> 20 in the feed-forward).
  // A wave's region of a ring slot is ALWAYS bytes [wave * 5120, wave * 5120 + 5120), whatever the tile kind: the 4-KiB quarter of a
  // GEMM-1 tile sits at the same 5-KiB stride (its image in memory stays packed).  With the quarters of the two tile kinds at their
  // natural offsets (w * 4096 vs w * 5120) a wave's DMA of a GEMM-1 quarter overwrote the tail of its neighbour's quarter of the full
  // tile that had the slot before - which the neighbour could still be reading, now that nothing holds the waves in step: one
  // generate() in four came out different (scripts/diag_race3.py; found by tests/test_gpu_models.py's bitwise-repeatability cases).
  auto piece = [&](int off, bool g1, bool live, int slot, int i) {
    const int j = wave * (g1 ? 4 : 5) + i;
    const bool on = live && !(g1 && i >= 4);
#ifdef TSD_CHAIN_NOPIECE  // ablation build: no DMA instructions at all after the first tiles of segment 1 (timing only)
    if (off >= SEG0_BYTES + 12 * TILE_FULL) return;
#endif
    blds16(on ? rw : rdead, lane16, (unsigned)(off + j * 1024), smem + RING_OFF + slot * SLOT + wave * 5120 + i * 1024);
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

  // ---- one GEMM stage: acc[4][5] (+)= Atile[64][nk*32] . W^T over the next nk tiles of the stream ------------------
  // EXTRA = VMEM loads the caller issued since the last DMA piece (bias / residual prefetch): they are younger than the
  // tiles the first wait is for.
  // Wave tile 64 x 80 (round 4; rounds 2-3 ran 2 x 2 waves of 32 x 160): 4 + 5 fragment reads per 20 MFMAs instead of 2 + 10 -
  // a tile step costs its MFMAs plus its fragment reads, added up (DESIGN.md 4.1 "what a K tile costs").
  const int a_rd = rsel * 128;
  const int w_rd = rsel * 64 + ((g ^ wswz(rsel)) << 4);
  auto gemm = [&](auto seg_c, auto zero_c, auto extra_c, f4 (&acc)[4][5], int a_base, int nk, auto swap_c) {
    constexpr int FN = 5, SEG = decltype(seg_c)::value, EXTRA = decltype(extra_c)::value;
    constexpr bool ZERO = decltype(zero_c)::value;
    // SWAP: D = Afrag x Wfrag^T instead of Wfrag x Afrag^T - a lane then holds ONE weight row (= lane & 15 of fragment b)
    // and FOUR consecutive token rows (4g + r of fragment a): the transposed output the V^T projection needs
    constexpr bool SWAP = decltype(swap_c)::value;
    if (ZERO) {
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // Software pipeline over the tiles (one wave per SIMD: nobody else hides the LDS latency): the fragment reads of
    // tile t are issued right after its barrier and land while the MFMAs of tile t-1 run from the other register set.
    h8 afA[4], wfA[FN], afB[4], wfB[FN];
    // One pipeline step = barrier of tile gt, then the 4*FN MFMAs of tile gt-1 (register set p*) with the 4 + FN fragment
    // reads of tile gt (into set n*) and the wave's DMA instructions for tile gt+4 spread between them.
    auto step = [&](int kt, h8 (&af)[4], h8 (&wf)[FN], const h8 (&paf)[4], const h8 (&pwf)[FN], bool have_prev, bool have_next) {
      bool g1 = false, live = false;
      int off = 0, s4 = 0;
      const char *sA = smem, *sW = smem;
      if (have_next) {
        tsd_jitter();
        // tile gt has landed: the three younger tiles (and the caller's EXTRA loads, which sit between tile gt+3 and tile
        // gt+4 of the stage's first tile in the queue) may stay in flight
        int younger = 0;
#pragma unroll
        for (int d = 1; d <= 3; d++) { bool yg1, ylive; tile_off(SEG, gt + d, yg1, ylive); younger += yg1 ? 4 : 5; }
        wait_alt_begin();  // exactly one of the four runs
        switch (younger) {
          case 12: if (EXTRA > 0 && kt < 4 && !TSD_CHAIN_STRICT_WAIT) wait_vm<12, EXTRA>(); else wait_vm<12>(); break;
          case 13: if (EXTRA > 0 && kt < 4 && !TSD_CHAIN_STRICT_WAIT) wait_vm<13, EXTRA>(); else wait_vm<13>(); break;
          case 14: if (EXTRA > 0 && kt < 4 && !TSD_CHAIN_STRICT_WAIT) wait_vm<14, EXTRA>(); else wait_vm<14>(); break;
          default: if (EXTRA > 0 && kt < 4 && !TSD_CHAIN_STRICT_WAIT) wait_vm<15, EXTRA>(); else wait_vm<15>(); break;
        }
        wait_alt_end();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's reads of tile gt-1 are complete: its slot may be refilled
        if (TSD_CHAIN_TILE_BARRIER & 4) __builtin_amdgcn_s_sleep(1);
        if (!have_prev || (TSD_CHAIN_TILE_BARRIER & 1)) __builtin_amdgcn_s_barrier();  // stage start: the A operand every wave wrote a part of is complete
        asm volatile("" ::: "memory");
        sA = smem + a_base + (kt >> 1) * A_KT + a_rd + ((((kt & 1) * 4 + g) ^ key) << 4);
        sW = smem + RING_OFF + sl * SLOT + wn * 5120 + w_rd;
        off = tile_off(SEG, gt + 4, g1, live);
        s4 = sl == 0 ? 4 : sl - 1;  // (sl + 4) % 5: the slot tile gt-1 just left
      }
      constexpr int NM = 4 * FN, NR = 4 + FN;
      if (!have_prev) {  // first tile of the stage: nothing to multiply yet
#pragma unroll
        for (int a = 0; a < 4; a++) af[a] = *(const h8*)(sA + a * 2048);
#pragma unroll
        for (int b = 0; b < FN; b++) wf[b] = *(const h8*)(sW + b * 1024);
#pragma unroll
        for (int i = 0; i < 5; i++) if (!(g1 && i == 4)) piece(off, g1, live, s4, i);
      } else {
#pragma unroll
        for (int q = 0; q < NM; q++) {
          const int b = q >> 2, a = q & 3;
          if (SWAP) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(paf[a], pwf[b], acc[a][b], 0, 0, 0);
          else acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[b], paf[a], acc[a][b], 0, 0, 0);
          if (have_next) {
            __builtin_amdgcn_sched_barrier(0);
            // reads: one after each of the first NR MFMAs ... (A fragments first: every MFMA of the next step needs one)
            if (q < NR) {
              if (q < 4) af[q] = *(const h8*)(sA + q * 2048);
              else wf[q - 4] = *(const h8*)(sW + (q - 4) * 1024);
            }
            // ... DMA: one after every fourth MFMA
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
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I25 = std::integral_constant<int, 25>;
  using Yes = std::true_type;
  using No = std::false_type;

  // ---- accumulator-layout helpers: lane (rsel, g) of fragment row a holds columns cbase + b*4 + r ------------------
  const int cbase = wn * 80 + g * 20;
  auto load_cols = [&](const float* vec, f4 (&bv)[5]) {
#pragma unroll
    for (int b = 0; b < 5; b++) bv[b] = *(const f4*)(vec + cbase + b * 4);
  };
  // fp16 rows (residual sources): 20 consecutive columns = five 8-B loads per fragment row
  auto load_rows_raw = [&](const half_t* src, int ld, h4 (&raw)[4][5]) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const half_t* rp = src + (long long)(m0 + a * 16 + rsel) * ld + cbase;
#pragma unroll
      for (int b = 0; b < 5; b++) raw[a][b] = *(const h4*)(rp + b * 4);
    }
  };
  // LDS byte offset, inside a swizzled [k-tile][64 rows][128 B] A-operand tile, of the 8-B piece holding columns col .. col+3 of `row`
  auto a_piece = [&](int row, int col) { return (col >> 6) * A_KT + row * 128 + ((((col >> 3) & 7) ^ (row & 7)) << 4) + (col & 4) * 2; };
  // this lane's five 8-byte pieces of fragment row 0 in the A tile: fragment row a is 16 rows = 2048 bytes further on and has the same
  // swizzle key, so a store is base + immediate (computed per store, the swizzle arithmetic was a tenth of a LayerNorm's instructions)
  int apo[5];
#pragma unroll
  for (int b = 0; b < 5; b++) apo[b] = A_OFF + a_piece(rsel, cbase + b * 4);
  // write fp16((v - sub) * mul) into the A tile (swizzled A-operand layout)
  auto store_a_tile = [&](const f4 (&v)[4][5], const float (&mul)[4], const float (&sub)[4]) {
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 5; b++) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) o[r] = (half_t)((v[a][b][r] - sub[a]) * mul[a]);
        *(h4*)(smem + apo[b] + a * 2048) = o;
      }
    }
  };
  // LayerNorm of the residual stream into the A tile: (x - mean) / (sigma + eps), population sigma, no affine
  // (helpers/utils.mojo:2052-2061 via :1845-1885; App.A D8).  Each column wave computes the exact two-pass mean / M2
  // of its 80 columns; the four are merged after ONE exchange (Chan et al., equal counts).
  float* scr = (float*)(smem + SCR_OFF);
  auto layernorm_to_a = [&](const f4 (&v)[4][5]) {
    float mw[4], m2w[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < 5; b++) t += (v[a][b][0] + v[a][b][1]) + (v[a][b][2] + v[a][b][3]);
      t = rows_sum(t);
      mw[a] = t * (1.f / 80.f);
      float u = 0.f;
#pragma unroll
      for (int b = 0; b < 5; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const float d = v[a][b][r] - mw[a]; u += d * d; }
      u = rows_sum(u);
      m2w[a] = u;
      if (g == 0) *(f2*)(scr + ((a * 16 + rsel) * 4 + wn) * 2) = f2{mw[a], u};
    }
    lds_barrier();  // also: every wave is past its last read of the old A tile
    float mean[4], rs[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const f4 p01 = *(const f4*)(scr + (a * 16 + rsel) * 8), p23 = *(const f4*)(scr + (a * 16 + rsel) * 8 + 4);
      const float mu = 0.25f * ((p01[0] + p01[2]) + (p23[0] + p23[2]));
      const float d0 = p01[0] - mu, d1 = p01[2] - mu, d2 = p23[0] - mu, d3 = p23[2] - mu;
      const float m2 = ((p01[1] + p01[3]) + (p23[1] + p23[3])) + 80.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      mean[a] = mu;
      rs[a] = 1.f / (sqrtf(m2 * (1.f / C)) + p.eps);
    }
    store_a_tile(v, rs, mean);
  };
  const float one4[4] = {1.f, 1.f, 1.f, 1.f}, zero4[4] = {0.f, 0.f, 0.f, 0.f};

  if constexpr (KIND == KIND_HEAD) {
    // =============================================================================================================
    // head of the block (diffusion.mojo:116-124): GroupNorm-apply -> conv_in (1x1) -> tok ; LayerNorm -> q, k, V^T
    f4 T[4][5], acc[4][5], accq[4][5], bv[5];
    h4 raw[4][5];
    load_rows_raw(p.x, p.ld_x, raw);
    f2 gst[2];  // (mean, 1/(sigma+eps)) of this lane's two groups of 10 channels
#pragma unroll
    for (int k = 0; k < 2; k++) gst[k] = *(const f2*)(p.gn_stats + ((long long)bsmp * 32 + wn * 8 + g * 2 + k) * 2);
    load_cols(p.b_in, bv);
    seg_begin(0);
    // GroupNorm (32 groups, eps 1e-6, no SiLU; helpers/utils.mojo:1845-1885) of the x tile -> A tile
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int row = a * 16 + rsel;
#pragma unroll
      for (int b = 0; b < 5; b++) {
        h4 o;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int grp = (b * 4 + r) / 10;
          o[r] = (half_t)(((float)raw[a][b][r] - gst[grp][0]) * gst[grp][1]);
        }
        *(h4*)(smem + apo[b] + a * 2048) = o;
      }
    }
    CTS(1);
    // ---- tok = GN(x) . Wc^T + b  (diffusion.mojo:117) ; the first tile barrier publishes the A tile -----------------------
    // No global store happens before the last GEMM is done: stores share vmcnt with the DMA stream, and draining them
    // (an HBM write round trip) in the middle of the kernel costs more than any stage.  tok / q / k wait in registers.
    gemm(I0{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
    h4 tokh[4][5];
#pragma unroll
    for (int a = 0; a < 4; a++) {
#pragma unroll
      for (int b = 0; b < 5; b++) {
        T[a][b] = acc[a][b] + bv[b];
#pragma unroll
        for (int r = 0; r < 4; r++) tokh[a][b][r] = (half_t)T[a][b][r];
      }
    }
    CTS(2);
    layernorm_to_a(T);   // the LayerNorm sees the fp32 tok (the unfused graph normalises its fp16 rounding)
    CTS(3);
    // ---- q, k = LN(tok) . W^T  (helpers/attention.mojo:29, in_bias = False) -----------------------------------------------
    f4 acck[4][5];
    gemm(I0{}, Yes{}, I0{}, accq, A_OFF, 10, No{});
    gemm(I0{}, Yes{}, I0{}, acck, A_OFF, 10, No{});
    CTS(4);
    CTS(5);
    // ---- V^T = Wv . LN(tok)^T : swapped operands, lane (channel wn*80 + b*16 + rsel) holds tokens a*16 + 4g + r ----
    gemm(I0{}, Yes{}, I0{}, acc, A_OFF, 10, Yes{});
    CTS(6);
    wait_vm<0>();   // the dead tail DMAs (zero-size descriptors still WRITE zeros) have landed: the ring is really free
    lds_barrier();  // every wave is done with the A tile and the ring: both become output staging
    // Outputs leave through LDS so that every global store instruction writes whole contiguous row segments (lane-owned
    // 40-B row pieces stored directly are 64 scattered 8-B writes per instruction: store-issue bound).
    //   A tile region : V^T [320 channels][64 tokens] (128-B rows, 16-B chunks XOR-swizzled by channel & 7)
    //   ring region   : two row-major [64 rows][640 B] tiles at a 656-B pitch
    constexpr int RP = 656, ST0 = RING_OFF, ST1 = RING_OFF + 64 * RP;
    static_assert(ST1 + 64 * RP <= SCR_OFF, "staging tiles must fit in the ring region");
    // stage fragment rows: this lane's 4 columns cbase + 4b .. of fragment row a
#define STAGE_ROWS_ACC(base, v)                                                                                         \
  _Pragma("unroll") for (int a = 0; a < 4; a++) _Pragma("unroll") for (int b = 0; b < 5; b++) {                          \
    h4 o_;                                                                                                              \
    _Pragma("unroll") for (int r = 0; r < 4; r++) o_[r] = (half_t)v[a][b][r];                                            \
    *(h4*)(smem + (base) + (a * 16 + rsel) * RP + (cbase + b * 4) * 2) = o_;                                             \
  }
    auto flush_rows = [&](int base, half_t* dst, int ld) {  // 2560 16-B chunks, 40 per row: a wave stores 1 KiB = 1.6 full rows
#pragma unroll
      for (int i = 0; i < 10; i++) {
        const int t = tid + 256 * i, row = t / 40, c = t - row * 40;
        *(h8*)(dst + (long long)(m0 + row) * ld + c * 8) = *(const h8*)(smem + base + row * RP + c * 16);
      }
    };
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) *(h4*)(smem + ST0 + (a * 16 + rsel) * RP + (cbase + b * 4) * 2) = tokh[a][b];
    STAGE_ROWS_ACC(ST1, accq)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) {
        const int c = wn * 80 + b * 16 + rsel;
        const int tkn = a * 16 + 4 * g;  // first of this lane's four tokens
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
  f4 T[4][5];     // residual stream (fp32)
  f4 acc[4][5];   // stage accumulators
  f4 bv[5];       // per-column epilogue vector of the current stage (prefetched under the stage's GEMM)
  h4 raw[4][5];   // residual rows in flight (tok, later x)
  // ---- prologue: ao tile -> A tile ; residual 1 and the first bias in flight ; four weight tiles ahead ---------------
  {
    const rsrc_t ra = make_rsrc(p.ao);
#pragma unroll
    for (int i = 0; i < 10; i++) {  // 40 pieces of [8 rows][128 B]: piece j = k-tile j/8, rows (j%8)*8 ..
      const int j = wave + 4 * i, kt = j >> 3, row = (j & 7) * 8 + lrow;
      blds16(ra, (unsigned)((m0 + row) * p.ld_ao + cch * 8) * 2, kt * 128, smem + A_OFF + kt * A_KT + (j & 7) * 1024);
    }
  }
  seg_begin(0);
  // residual 1 and the first bias are needed only after the first GEMM: issued BEHIND the ao tile and the first four weight tiles,
  // so that GEMM starts as soon as those have landed (its first four tile waits leave these 25 loads in flight) - the 20 row loads
  // are 8-byte pieces of 16 rows each, slow through the texture-address unit, and used to sit in front of the first tile wait
  load_rows_raw(p.tok, p.ld_tok, raw);
  load_cols(p.bso, bv);
  CTS(1);
  // ---- tok2 = ao . Wso^T + b + tok ------------------------------------------------------------------------------------
  gemm(I0{}, Yes{}, I25{}, acc, A_OFF, 10, No{});
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 5; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) T[a][b][r] = (float)raw[a][b][r] + (acc[a][b][r] + bv[b][r]);
  CTS(2);
  layernorm_to_a(T);
  CTS(3);
  // ---- q = LN(tok2) . Wq^T  (scaled by softmax scale * log2 e) -----------------------------------------------------------
  gemm(I0{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
  CTS(4);
  wait_vm<0>();   // the dead tail of segment 0 has landed: the ring region is free for the context keys / values
  lds_barrier();  // every wave is done reading LN(tok2) and the last weight tile
  CTS(10);
  // ---- cross attention over the sample's T context keys (helpers/attention.mojo:105-115) ------------------------------
  // Ring region: K tile 5 k-tiles x [80 key rows][128 B] (weight-operand layout) | V^T keys 0..63 [320 rows][128 B] |
  // V^T keys 64..79 [320 rows][32 B].  K row b*16 + ii holds key 32*(b>>1) + 8*(ii>>2) + 4*(b&1) + (ii&3) for b < 4
  // (output lane (g, r) of the key fragments 2kk, 2kk+1 then holds keys 32kk + 8g + 0..7: the A fragment of P.V k-step
  // kk) and key 64 + ii for b = 4 (k-step 2 pairs lane group g with keys 64 + 4g .. + 3).
  constexpr int KT_B = 80 * 128, V0_OFF = 5 * KT_B, V1_OFF = V0_OFF + 320 * 128;
  static_assert(V1_OFF + 320 * 32 <= NSLOT * SLOT, "context tiles must fit in the ring region");
  {
    // 100 pieces of 1 KiB, 25 per wave.  Round 4: the per-lane part of every address is computed ONCE (the K row-block offsets of
    // this wave's row blocks, one V^T offset per region, masks folded in) and the piece number only moves the scalar offset and the
    // LDS destination - the round-2 loop recomputed key permutation, row and mask per piece behind wave-uniform branches and
    // exec-masked selects: 6.8 K ticks of issue for 25 pieces (profiles/r04_chain_ticks_*.txt).
    const rsrc_t rk = make_rsrc(p.Kc + bsmp * p.sK), rv = make_rsrc(p.Vt + bsmp * p.sVt);
    const int nch = (p.T + 7) >> 3;
    // K: row block rb (8 LDS rows) of every k-tile; this wave's row blocks are rb = wave, wave + 4, wave + 8
#pragma unroll
    for (int m = 0; m < 3; m++) {
      const int rb = wave + 4 * m;  // wave-uniform
      if (rb < 10) {
        const int rho = rb * 8 + lrow, b = rho >> 4, ii = rho & 15;
        const int kidx = b < 4 ? 32 * (b >> 1) + 8 * (ii >> 2) + 4 * (b & 1) + (ii & 3) : 64 + ii;
        const unsigned live = (unsigned)((kidx - p.T) >> 31);  // all ones for a valid key
        const unsigned koff = (((unsigned)(kidx * p.ldk + cch * 8) * 2) & live) | (PAD_OFF & ~live);
#pragma unroll
        for (int kt = 0; kt < 5; kt++) blds16(rk, koff, kt * 128, smem + RING_OFF + kt * KT_B + rb * 1024);
      }
    }
    // V^T keys 0..63: piece j' = rows 8j' .. 8j'+7 (40 pieces: waves 0, 1 take 8 each, waves 2, 3 twelve); keys 64..79: piece j'' = rows
    // 32j'' .. (10 pieces: 2, 2, 3, 3) - 15 + 8 + 2 = 10 + 12 + 3 = 25 pieces per wave
    const unsigned vlive0 = (unsigned)((cch - nch) >> 31), vlive1 = (unsigned)((8 + (lane & 1) - nch) >> 31);
    const unsigned v0off = (((unsigned)(lrow * p.ldvt + cch * 8) * 2) & vlive0) | (PAD_OFF & ~vlive0);
    const unsigned v1off = (((unsigned)((lane >> 1) * p.ldvt + (8 + (lane & 1)) * 8) * 2) & vlive1) | (PAD_OFF & ~vlive1);
    const int v0s = wave < 2 ? wave * 8 : 16 + (wave - 2) * 12, v0n = wave < 2 ? 8 : 12;
    const int v1s = wave < 2 ? wave * 2 : 4 + (wave - 2) * 3, v1n = wave < 2 ? 2 : 3;
#pragma unroll
    for (int q = 0; q < 12; q++)
      if (q < v0n) blds16(rv, v0off, (unsigned)((v0s + q) * 8 * p.ldvt) * 2, smem + RING_OFF + V0_OFF + (v0s + q) * 1024);
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (q < v1n) blds16(rv, v1off, (unsigned)((v1s + q) * 32 * p.ldvt) * 2, smem + RING_OFF + V1_OFF + (v1s + q) * 1024);
  }
  {
    const float qs4[4] = {p.qscale, p.qscale, p.qscale, p.qscale};
    store_a_tile(acc, qs4, zero4);  // q -> A tile while the context tiles fly
  }
  load_cols(p.bco, bv);
  CTS(11);
  wait_vm<0, TSD_CHAIN_STRICT_WAIT ? 0 : 5>();   // the context tiles have landed (the 5 bias loads are younger)
  lds_barrier();  // q tile and context tiles visible
  CTS(12);
  // per wave: all 64 rows, heads 2*wn and 2*wn+1 (its own 80 columns of q: no other wave reads or writes them)
  const bool t_ge64 = p.T >= 64;  // wave-uniform: only key fragment 4 (keys 64 + 4g + r) can hold a key >= T
  f4 mk4;
#pragma unroll
  for (int r = 0; r < 4; r++) mk4[r] = 64 + 4 * g + r >= p.T ? -1.0e30f : 0.f;
#pragma unroll
  for (int hh = 0; hh < 2; hh++) {
    const int h = wn * 2 + hh, c0 = h * 5;  // first 16-B chunk of the head's 40 columns
    f4 sc[4][5];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
      for (int b = 0; b < 5; b++) sc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 2; st++) {
      // k-step 0: chunks c0 .. c0+3 ; k-step 1: chunk c0+4 in lane group 0, zeros in the key operand elsewhere
      const int cg = st == 0 ? c0 + g : c0 + 4;
      const int ktile = cg >> 3, cpos = cg & 7;
      h8 qa[4], kb[5];
#pragma unroll
      for (int a = 0; a < 4; a++) qa[a] = *(const h8*)(smem + A_OFF + ktile * A_KT + a_rd + a * 2048 + ((cpos ^ key) << 4));
#pragma unroll
      for (int b = 0; b < 5; b++) {
        kb[b] = *(const h8*)(smem + RING_OFF + ktile * KT_B + (b * 16 + rsel) * 128 + ((cpos ^ key) << 4));
        if (st == 1 && g != 0) kb[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 5; b++) sc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb[b], qa[a], sc[a][b], 0, 0, 0);
    }
    // softmax over the keys of each query row: 20 in-lane scores x 4 lane groups (max subtraction, App.A D6).
    // Round 4, after reading the ISA (the phase was ~3000 VALU instructions per wave for 152 MFMAs): (i) keys >= T are masked by
    // ADDING -1e30 to the one fragment that can hold them when T >= 64 (every other fragment only in the rare T < 64 case) instead of
    // a compare + select per score; (ii) the row sum comes out of the P.V MFMAs - V^T row 40 of the head's third 16-row fragment
    // (a discarded output channel) is replaced by ones, so output channel 40 is the sum of the SAME fp16-rounded probabilities the
    // MFMA multiplies (as flash_attn_kernel's ones row) - instead of a convert-back and an add per probability.
    h8 pf[4][3];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      if (!t_ge64) {
#pragma unroll
        for (int b = 0; b < 4; b++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (32 * (b >> 1) + 8 * g + 4 * (b & 1) + r >= p.T) sc[a][b][r] = -1.0e30f;
      }
      sc[a][4] += mk4;
      float mx = sc[a][0][0];
#pragma unroll
      for (int b = 0; b < 5; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) mx = fmaxf(mx, sc[a][b][r]);
      mx = rows_max(mx);
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        h8 o = h8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < (kk < 2 ? 8 : 4); j++) o[j] = (half_t)__builtin_amdgcn_exp2f(sc[a][kk < 2 ? 2 * kk + (j >> 2) : 4][j & 3] - mx);
        pf[a][kk] = o;
      }
    }
    // O_h = P . V_h : channel rows h*40 + b*16 + rsel of the V^T tiles (rows past the head's 40 give discarded outputs - except
    // row 40, which is the ones row: lane group 2, register 0 of fragment 2 then holds the row sum)
    f4 oc[4][3];
#pragma unroll
    for (int a = 0; a < 4; a++)
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
      if (rsel == 8) vb[2] = kk < 2 ? h8{1, 1, 1, 1, 1, 1, 1, 1} : h8{1, 1, 1, 1, 0, 0, 0, 0};
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) oc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[b], pf[a][kk], oc[a][b], 0, 0, 0);
    }
    float rinv[4];
#pragma unroll
    for (int a = 0; a < 4; a++) rinv[a] = 1.f / rows_sum(g == 2 ? oc[a][2][0] : 0.f);
    if (hh == 0) CTS(13);
    // The head's output overwrites the head's q columns in the A tile.  Columns h*40 .. h*40+39 are read (as q) and written by
    // THIS wave only, and its QK^T for this head is done, so no barrier is needed.  lane (g, r) of fragment b holds channel
    // h*40 + b*16 + 4g + r.
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int row = a * 16 + rsel;
#pragma unroll
      for (int b = 0; b < 3; b++) {
        const int d = b * 16 + 4 * g;
        if (d < 40) {
          h4 o;
#pragma unroll
          for (int r = 0; r < 4; r++) o[r] = (half_t)(oc[a][b][r] * rinv[a]);
          *(h4*)(smem + A_OFF + a_piece(row, h * 40 + d)) = o;
        }
      }
    }
  }
  lds_barrier();  // attention output complete in the A tile ; the ring region is free again
  CTS(5);
  seg_begin(1);
  // ---- tok3 = attn . Wco^T + b + tok2 -------------------------------------------------------------------------------
  gemm(I1{}, Yes{}, I0{}, acc, A_OFF, 10, No{});
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 5; b++) T[a][b] += acc[a][b] + bv[b];
  layernorm_to_a(T);
  CTS(6);
  // ---- GEGLU feed-forward: ten chunks of 128 hidden units; the second GEMM accumulates over the chunks ------------------
  f4 acc2[4][5];
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 5; b++) acc2[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  {
    // ONE software pipeline over the 140 GEGLU tiles (round 3).  Round 2 ran per chunk GEMM-1 (10 tiles) -> a * gelu(g) ->
    // GEMM-2 (4 tiles) as three serial phases.  Here a step = barrier of tile gt, the MFMAs of tile gt-1 (whatever GEMM it belongs
    // to) from one fragment register set, the fragment reads of tile gt into the other, the DMA of tile gt+4 - straight through the walk
    //   G1(0) | G1(1) G2(0) | G1(2) G2(1) | ... | G1(9) G2(8) | G2(9)
    // and chunk j's activation is computed under the MFMAs of G2(j-1)'s tiles (the accumulator of G1(j) is complete by then and
    // G1(j+1) has not started), kept in 16 registers, and written to the activation tile at the second step of G1(j+1) - two
    // barriers after G2(j-1)'s last read of that tile, nine steps before G2(j)'s first.
    // Round 4, from the ISA of the round-3 loop: (i) the activation arithmetic has no side effects, so nothing tied it to the place
    // it was written at - hipcc sank all 32 activations of a chunk (~500 VALU instructions, 32 v_exp + 32 v_rcp) into the ONE step
    // that stores them, where they ran with the matrix pipe idle: every activation pair now ends in an `asm volatile` use of its
    // packed result, which pins it between the MFMAs it was meant to hide under; (ii) the walk is STATIC - a step knows at compile
    // time which tile is four ahead and how many DMA pieces are in flight, so the ~40 scalar instructions (and the branches) per step
    // that recomputed this from the tile counter are gone; (iii) the GEGLU-1 bias enters as the C operand of the chunk's first
    // MFMAs instead of two v_add per activation.
    // GEMM-1 tile: 256 interleaved (a, g) columns = 64 per wave (FN = 4): a lane's 16 consecutive columns are 8 (a, g) pairs = the
    // 8 activations of hidden units wn*32 + g*8 .. +7 of one fragment row - one 16-B chunk of the activation tile.
    f4 b1v[4];
    unsigned oreg[4][4];  // [fragment row][pair]: two fp16 activations each
    h8 afS[2][4], wfS[2][5];
    // GEMM-1's A operand - LN(tok3), the same for all ten chunks - keeps the fragments of its first NRES k-tiles in registers for the
    // whole feed-forward: those steps read 4 weight fragments instead of 4 + 4 (a ds_read_b128 holds a one-wave SIMD's issue ~30 cycles)
    constexpr int NRES = TSD_CHAIN_ARES;
    h8 ares[NRES > 0 ? NRES : 1][4];
    if constexpr (NRES > 0) {
      lds_barrier();  // LN(tok3) complete in the A tile (all four waves' columns)
#pragma unroll
      for (int kc = 0; kc < NRES; kc++)
#pragma unroll
        for (int a = 0; a < 4; a++)
          ares[kc][a] = *(const h8*)(smem + A_OFF + (kc >> 1) * A_KT + a_rd + a * 2048 + ((((kc & 1) * 4 + g) ^ key) << 4));
    }
    auto load_b1 = [&](int jc) {
      const float* bp = p.b1 + jc * 256 + wn * 64 + g * 16;
#pragma unroll
      for (int b = 0; b < 4; b++) b1v[b] = *(const f4*)(bp + b * 4);
    };
    // activations 2*jp and 2*jp+1 of fragment row u: hidden units (a, g) = (acc[u][jp][0], [1]) and ([2], [3]); the bias is already
    // in the accumulator.  The asm use keeps the arithmetic HERE (see (i) above).
    auto act_pair = [&](int u, int jp) {
      const f4 v = acc[u][jp];
      const float o0 = v[0] * gelu_tanh_c(v[1]), o1 = v[2] * gelu_tanh_c(v[3]);
      h2 pk = {(half_t)o0, (half_t)o1};
      oreg[u][jp] = pin_here(__builtin_bit_cast(unsigned, pk));
    };
    auto write_act = [&]() {
      typedef unsigned u4v __attribute__((ext_vector_type(4)));
#pragma unroll
      for (int a = 0; a < 4; a++)
        *(u4v*)(smem + ACT_OFF + (wn >> 1) * A_KT + (a * 16 + rsel) * 128 + ((((wn & 1) * 4 + g) ^ key) << 4)) =
            u4v{oreg[a][0], oreg[a][1], oreg[a][2], oreg[a][3]};
    };
    constexpr int F_BIAS = 1, F_EXTRA = 2, F_WRITE = 4, F_PLAIN = 8, F_SYNC = 16;
    constexpr int FB = SEG0_BYTES + 10 * TILE_FULL;  // byte offset of chunk 0's tiles in the stream
    int jb = FB;                                      // ... of the current chunk j's
    // PK / CK: kind of the previous / current tile (0 none, 1 = GEMM-1 tile: 256 rows, FN 4, A from the LN tile ; 2 = GEMM-2
    // tile: 320 rows, FN 5, A from the activation tile) ; KC: index of the current tile in its GEMM ; SET: fragment register
    // set the current tile's fragments go to ; FILL: activation unit computed under this step's MFMAs (-1 none)
    // LK / LOFF: the tile four ahead (the one this step's DMA pieces fetch): kind (1 / 2 as above, 0 = none) and byte offset relative to jb
    // YG: pieces of the three tiles behind the current one that may stay in flight (4 per GEMM-1 tile, 5 per full tile)
    auto gstep = [&](auto pk_c, auto ck_c, auto kc_c, auto set_c, auto fill_c, auto flags_c, auto lk_c, auto loff_c, auto yg_c) {
      constexpr int PK = decltype(pk_c)::value, CK = decltype(ck_c)::value, KC = decltype(kc_c)::value, SET = decltype(set_c)::value;
      constexpr int FILL = decltype(fill_c)::value, FLAGS = decltype(flags_c)::value;
      constexpr int LK = decltype(lk_c)::value, LOFF = decltype(loff_c)::value, YG = decltype(yg_c)::value;
      constexpr int FNp = PK == 1 ? 4 : 5, FNc = CK == 1 ? 4 : 5;
      constexpr bool G1N = LK == 1;
      int s4 = 0;
      const char *sA = smem, *sW = smem;
      if constexpr (CK != 0) {
        tsd_jitter();
        constexpr int EX = ((FLAGS & F_EXTRA) && !TSD_CHAIN_STRICT_WAIT) ? 4 : 0;
        if constexpr (!(TSD_CHAIN_ABL & 32)) wait_vm<YG, EX>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads of tile gt-1 (and any activation-tile writes) are complete
        // barriers only where the shared activation tile changes hands: before it is rewritten (F_WRITE steps: every wave has finished
        // the previous chunk's GEMM-2 reads) and before its first GEMM-2 read (F_SYNC: every wave's part is written)
        if constexpr ((TSD_CHAIN_TILE_BARRIER & 4) != 0) __builtin_amdgcn_s_sleep(1);
        if constexpr (((FLAGS & (F_WRITE | F_SYNC)) != 0 || (TSD_CHAIN_TILE_BARRIER & 2)) && !(TSD_CHAIN_ABL & 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        sA = smem + (CK == 1 ? A_OFF : ACT_OFF) + (KC >> 1) * A_KT + a_rd + ((((KC & 1) * 4 + g) ^ key) << 4);
        sW = smem + RING_OFF + sl * SLOT + wn * 5120 + w_rd;
        s4 = sl == 0 ? 4 : sl - 1;
      }
      const int off = jb + LOFF;
      auto dma = [&](int i) {  // piece i of the tile four ahead (a GEMM-1 tile has 16 pieces: 4 per wave)
        if constexpr (CK != 0 && LK != 0 && !(TSD_CHAIN_ABL & 8)) { if (!(G1N && i == 4)) piece(off, G1N, true, s4, i); }
      };
      if constexpr ((FLAGS & F_WRITE) != 0) write_act();
      // register-resident A fragments: the current tile's are not read (CRES), the previous tile's come from ares (PRES)
      constexpr int PKC = (CK == 1 && KC > 0) ? KC - 1 : 9;  // index of the previous tile when it is a GEMM-1 tile
      constexpr bool CRES = CK == 1 && KC < NRES, PRES = PK == 1 && PKC < NRES;
      h8 (&af)[4] = afS[SET];
      h8 (&wf)[5] = wfS[SET];
      const h8 (&paf)[4] = PRES ? ares[PKC < NRES ? PKC : 0] : afS[SET ^ 1];
      const h8 (&pwf)[5] = wfS[SET ^ 1];
      if constexpr (PK == 0) {
        if constexpr (CK != 0) {
#pragma unroll
          for (int a = 0; a < 4; a++) if (!CRES) af[a] = *(const h8*)(sA + a * 2048);
#pragma unroll
          for (int b = 0; b < FNc; b++) wf[b] = *(const h8*)(sW + b * 1024);
#pragma unroll
          for (int i = 0; i < 5; i++) dma(i);
        }
      } else {
        constexpr int NA = CRES ? 0 : 4;  // A fragments to read for the current tile
        constexpr int NM = 4 * FNp, NR = CK != 0 ? NA + FNc : 0;
#pragma unroll
        for (int q = 0; q < NM; q++) {
          const int b = q >> 2, a = q & 3;
          if constexpr (TSD_CHAIN_ABL & 4) {
            asm volatile("" ::"v"(pwf[b]), "v"(paf[a]));
          } else if constexpr (PK == 1) {
            const f4 c0 = (FLAGS & F_BIAS) ? b1v[b] : acc[a][b];  // first products of a chunk: the GEGLU-1 bias is the C operand
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[b], paf[a], c0, 0, 0, 0);
          } else {
            acc2[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pwf[b], paf[a], acc2[a][b], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (q < NR && !(TSD_CHAIN_ABL & 2)) {
            if (q < NA) af[q] = *(const h8*)(sA + q * 2048);
            else wf[q - NA] = *(const h8*)(sW + (q - NA) * 1024);
          }
          if ((q + 1) % (NM / 5) == 0 && (q + 1) / (NM / 5) - 1 < 5) dma((q + 1) / (NM / 5) - 1);
          // one activation pair after every fifth MFMA (4 per step = one fragment row)
          if (FILL >= 0 && NM == 20 && q % 5 == 4 && !(TSD_CHAIN_ABL & 16)) act_pair(FILL, q / 5);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if constexpr ((FLAGS & F_PLAIN) != 0) {  // chunk 0: nothing to hide its activation under
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int jp = 0; jp < 4; jp++) act_pair(u, jp);
        write_act();
      }
      if constexpr (CK != 0) {
        gt++;
        sl = sl == 4 ? 0 : sl + 1;
      }
    };
    // compile-time description of the walk.  Position P of a chunk iteration j (1..9): P < 10 = tile P of G1(j), else tile P-10 of
    // G2(j-1).  The tile D positions ahead: same iteration while P + D < 14, else the next iteration's G1(j+1) (LAST: G2(9)).
    // Offsets are relative to jb = the byte offset of chunk j.
#define TSD_IC(x) std::integral_constant<int, (x)>{}
    auto la_kind = [](int pd, bool last) constexpr { return pd < 10 ? 1 : (pd < 14 ? 2 : (last ? 2 : 1)); };
    auto la_off = [](int pd, bool last) constexpr {
      return pd < 10 ? pd * TILE_G1
                     : (pd < 14 ? -FFN_CHUNK_BYTES + 10 * TILE_G1 + (pd - 10) * TILE_FULL
                                : (last ? 10 * TILE_G1 + (pd - 14) * TILE_FULL : FFN_CHUNK_BYTES + (pd - 14) * TILE_G1));
    };
    auto la_yg = [=](int p_, bool last) constexpr {
      int y = 0;
      for (int d = 1; d <= 3; d++) y += la_kind(p_ + d, last) == 1 ? 4 : 5;
      return y;
    };
    // one chunk iteration: FIRST (j = 1: the previous tile is G1(0)'s last, chunk 0's activation runs in the open), LAST (j = 9)
    auto iteration = [&](auto first_c, auto last_c) {
      constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
#define TSD_LA(P) TSD_IC(la_kind((P) + 4, LAST)), TSD_IC(la_off((P) + 4, LAST)), TSD_IC(la_yg((P), LAST))
      // ---- G1(j): the first step finishes the previous tile (G1(0)'s last for j = 1; else G2(j-2)'s last, hiding fragment row 3 of
      // chunk j-1's activation) ; the second publishes chunk j-1's activations ----
      if constexpr (FIRST) gstep(TSD_IC(1), TSD_IC(1), TSD_IC(0), TSD_IC(0), TSD_IC(-1), TSD_IC(F_PLAIN), TSD_LA(0));
      else gstep(TSD_IC(2), TSD_IC(1), TSD_IC(0), TSD_IC(0), TSD_IC(3), TSD_IC(0), TSD_LA(0));
      gstep(TSD_IC(1), TSD_IC(1), TSD_IC(1), TSD_IC(1), TSD_IC(-1), TSD_IC(FIRST ? F_BIAS : F_BIAS | F_WRITE), TSD_LA(1));
      gstep(TSD_IC(1), TSD_IC(1), TSD_IC(2), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA(2));
      gstep(TSD_IC(1), TSD_IC(1), TSD_IC(3), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA(3));
      gstep(TSD_IC(1), TSD_IC(1), TSD_IC(4), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA(4));
      if constexpr (!LAST) {
        load_b1((jb - FB) / FFN_CHUNK_BYTES + 1);                // the next chunk's bias: its C operand five steps into the next iteration
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(5), TSD_IC(1), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA(5));  // EXTRA for four steps
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(6), TSD_IC(0), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA(6));
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(7), TSD_IC(1), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA(7));
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(8), TSD_IC(0), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA(8));
      } else {
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(5), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA(5));
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(6), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA(6));
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(7), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA(7));
        gstep(TSD_IC(1), TSD_IC(1), TSD_IC(8), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA(8));
      }
      gstep(TSD_IC(1), TSD_IC(1), TSD_IC(9), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA(9));
      // ---- G2(j-1): chunk j's activation, fragment rows 0..2, under its tiles' MFMAs ----
      gstep(TSD_IC(1), TSD_IC(2), TSD_IC(0), TSD_IC(0), TSD_IC(-1), TSD_IC(F_SYNC), TSD_LA(10));
      gstep(TSD_IC(2), TSD_IC(2), TSD_IC(1), TSD_IC(1), TSD_IC(0), TSD_IC(0), TSD_LA(11));
      gstep(TSD_IC(2), TSD_IC(2), TSD_IC(2), TSD_IC(0), TSD_IC(1), TSD_IC(0), TSD_LA(12));
      gstep(TSD_IC(2), TSD_IC(2), TSD_IC(3), TSD_IC(1), TSD_IC(2), TSD_IC(0), TSD_LA(13));
#undef TSD_LA
    };
    // ---- G1(0): tiles 0..9 (bias of chunk 0 in flight under its first steps; the tile four ahead is G1(0)'s own, then G1(1)'s) ----
    load_b1(0);
#define TSD_LA0(P) TSD_IC(1), TSD_IC((P) + 4 < 10 ? ((P) + 4) * TILE_G1 : FFN_CHUNK_BYTES + ((P) + 4 - 10) * TILE_G1), TSD_IC(12)
    gstep(TSD_IC(0), TSD_IC(1), TSD_IC(0), TSD_IC(0), TSD_IC(-1), TSD_IC(NRES > 0 ? F_EXTRA : F_EXTRA | F_SYNC), TSD_LA0(0));  // (the ares block's barrier published LN(tok3))
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(1), TSD_IC(1), TSD_IC(-1), TSD_IC(F_BIAS), TSD_LA0(1));   // its MFMAs read b1v: the compiler waits for it
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(2), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA0(2));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(3), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA0(3));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(4), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_LA0(4));
    load_b1(1);
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(5), TSD_IC(1), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA0(5));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(6), TSD_IC(0), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA0(6));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(7), TSD_IC(1), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA0(7));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(8), TSD_IC(0), TSD_IC(-1), TSD_IC(F_EXTRA), TSD_LA0(8));
    gstep(TSD_IC(1), TSD_IC(1), TSD_IC(9), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_LA0(9));
#undef TSD_LA0
    jb = FB + FFN_CHUNK_BYTES;
    iteration(std::true_type{}, std::false_type{});
    for (int j = 2; j < 9; j++) {
      jb += FFN_CHUNK_BYTES;
      iteration(std::false_type{}, std::false_type{});
    }
    jb += FFN_CHUNK_BYTES;
    iteration(std::false_type{}, std::true_type{});
    // ---- G2(8)'s last tile with fragment row 3 of chunk 9's activation, publish, then G2(9); four ahead: the conv_out tiles ----
    // (offsets relative to jb = chunk 9: the Wout tiles follow chunk 9's fourteen)
    gstep(TSD_IC(2), TSD_IC(0), TSD_IC(0), TSD_IC(0), TSD_IC(3), TSD_IC(0), TSD_IC(0), TSD_IC(0), TSD_IC(0));
    // that step had no tile barrier (CK = 0): G2(8)'s last reads of the activation tile by the OTHER waves must have retired
    // before this wave overwrites its columns with chunk 9 (inside the loop the publish sits two barriers behind the last read)
    lds_barrier();
    write_act();
    gstep(TSD_IC(0), TSD_IC(2), TSD_IC(0), TSD_IC(0), TSD_IC(-1), TSD_IC(F_SYNC), TSD_IC(2), TSD_IC(FFN_CHUNK_BYTES + 0 * TILE_FULL), TSD_IC(15));
    gstep(TSD_IC(2), TSD_IC(2), TSD_IC(1), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_IC(2), TSD_IC(FFN_CHUNK_BYTES + 1 * TILE_FULL), TSD_IC(15));
    gstep(TSD_IC(2), TSD_IC(2), TSD_IC(2), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_IC(2), TSD_IC(FFN_CHUNK_BYTES + 2 * TILE_FULL), TSD_IC(15));
    gstep(TSD_IC(2), TSD_IC(2), TSD_IC(3), TSD_IC(1), TSD_IC(-1), TSD_IC(0), TSD_IC(2), TSD_IC(FFN_CHUNK_BYTES + 3 * TILE_FULL), TSD_IC(15));
    gstep(TSD_IC(2), TSD_IC(0), TSD_IC(0), TSD_IC(0), TSD_IC(-1), TSD_IC(0), TSD_IC(0), TSD_IC(0), TSD_IC(0));
#undef TSD_IC
  }
  CTS(7);
  load_cols(p.b2, bv);
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int b = 0; b < 5; b++) T[a][b] += acc2[a][b] + bv[b];
  // tok4 -> A tile (every wave passed the last GEGLU-1 tile long ago: the A tile is free)
  store_a_tile(T, one4, zero4);
  // ---- out = tok4 . Wout^T + b + x --------------------------------------------------------------------------------
  load_cols(p.bout, bv);
  load_rows_raw(p.x, p.ld_x, raw);
  gemm(I1{}, Yes{}, I25{}, acc, A_OFF, 10, No{});
  CTS(8);
  float gs1[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, gs2[2][2] = {{0.f, 0.f}, {0.f, 0.f}};  // [32-row slab][this lane's 2 groups of 10 channels]
  lds_barrier();  // every wave is done with the A tile: it (and the idle activation tile behind it) stages the output rows
  constexpr int RP = 656;  // row pitch of the row-major staging tile, whole-row global stores
  static_assert(A_OFF + 64 * RP <= RING_OFF, "output staging must stay clear of the ring (dead DMAs may still land there)");
#pragma unroll
  for (int a = 0; a < 4; a++) {
#pragma unroll
    for (int b = 0; b < 5; b++) {
      h4 o;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        o[r] = (half_t)((float)raw[a][b][r] + (acc[a][b][r] + bv[b][r]));
        const float f = (float)o[r];
        const int grp = (b * 4 + r) / 10;
        gs1[a >> 1][grp] += f; gs2[a >> 1][grp] += f * f;
      }
      *(h4*)(smem + A_OFF + (a * 16 + rsel) * RP + (cbase + b * 4) * 2) = o;
    }
  }
  lds_barrier();
#pragma unroll
  for (int i = 0; i < 10; i++) {  // 2560 16-B chunks, 40 per row: a wave instruction stores 1 KiB of consecutive row bytes
    const int t = tid + 256 * i, row = t / 40, c = t - row * 40;
    *(h8*)(p.out + (long long)(m0 + row) * p.ld_out + c * 8) = *(const h8*)(smem + A_OFF + row * RP + c * 16);
  }
  if (p.gn_part) {
    // GroupNorm(32) statistics of the rounded output for the consumer (one 32-row slab per pair of fragment rows)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int k = 0; k < 2; k++) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gs1[s][k] += __shfl_xor(gs1[s][k], o); gs2[s][k] += __shfl_xor(gs2[s][k], o); }
      }
    if (rsel == 0) {
#pragma unroll
      for (int s = 0; s < 2; s++) {
        const int mrow = m0 + s * 32, bb = mrow / p.S, slab = (mrow - bb * p.S) >> 5;
        float* ob = p.gn_part + (((long long)bb * p.gn_nslab + slab) * 32 + wn * 8 + g * 2) * 2;
        *(f4*)ob = f4{gs1[s][0], gs2[s][0], gs1[s][1], gs2[s][1]};
      }
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
  return ctx_set_option(ctx, &TsdOptions::chain, on ? 1 : 0, 0, 1);
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
