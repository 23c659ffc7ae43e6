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
//   * the weights of all stages form ONE stream of [160 rows][64 k] half-tiles DMA'd (buffer_load ... lds) through a
//     5-slot LDS ring, three half-tiles ahead of the MFMAs and across stage boundaries, behind counted s_waitcnt vmcnt;
//   * the 77 context keys / values of the sample (projected once per step) are staged in the ring region; the scores
//     never leave registers: swapped-operand QK^T leaves each lane with the scores of ONE query row, and the key -> MFMA
//     row permutation makes the fp16 probabilities the A fragments of the P.V MFMAs without any data movement;
//   * LayerNorm statistics are an in-lane sum + two lane shuffles + one LDS exchange between the two column waves;
//   * the output's GroupNorm statistics (for the next residual block) come from the rounded values in registers.
// MFMA: v_mfma_f32_16x16x32_f16 with swapped operands (D = Wfrag x Afrag^T), 4 waves as 2(M) x 2(N), a wave tile is
// 32 rows x 160 columns (FM = 2, FN = 10), one wave per SIMD, pinned issue order as in kernels_gemm.hip.
#include <math.h>
#include <stdlib.h>
#include <type_traits>

#include "common.h"
#include "lds_dma.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct TailK {
  const half_t* ao; const half_t* tok; const half_t* x; half_t* out;
  int ld_ao, ld_tok, ld_x, ld_out;
  const half_t *Wso, *Wq, *Wco, *W1, *W2, *Wout;
  const float *bso, *bco, *b1, *b2, *bout;
  const half_t* Kc; int ldk; long long sK;    // projected context keys   [B][>=T rows][ldk], this block's columns
  const half_t* Vt; int ldvt; long long sVt;  // projected context values [B][C rows][ldvt]  (key-contiguous)
  int T;                                      // valid keys (<= 96)
  int M, S;                                   // token rows, rows per sample (S % 64 == 0)
  float qscale;                               // softmax scale * log2(e), folded into q
  float eps;
  float* gn_part; int gn_nslab;               // output GroupNorm(32) statistics [B][nslab][32][2] (nullptr: none)
};

namespace {
constexpr int C = 320, BM = 64;
constexpr int A_OFF = 0, A_KT = 8192;                 // A tile: 5 k-tiles x [64 rows][128 B]
constexpr int ACT_OFF = 40960;                        // GEGLU activations: 2 k-tiles x [64][128 B]
constexpr int RING_OFF = ACT_OFF + 16384, HSLOT = 20480;  // 5 half-slots of [160 rows][128 B]
constexpr int SCR_OFF = RING_OFF + 5 * HSLOT;         // LayerNorm exchange: [2][64 rows][2 column waves] floats
constexpr int LDS_BYTES = SCR_OFF + 1024;
static_assert(LDS_BYTES <= 163840, "LDS budget");

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
}  // namespace

__global__ __launch_bounds__(256, 1) void attn_tail_kernel(const TailK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int rsel = lane & 15, key = lane & 7, g = lane >> 4;
  const int lrow = lane >> 3, cch = (lane & 7) ^ lrow;  // DMA: LDS row within an 8-row piece, logical 16-B chunk fetched
  const int m0 = blockIdx.x * BM;
  const int bsmp = m0 / p.S;

  // ---- per-lane DMA offsets of the three weight-tile shapes ------------------------------------------------------
  // LDS row rho of a half-tile holds weight row n(rho) such that output lane (g, r) of fragment b is column
  // g*4*FN + b*4 + r of the half: every lane ends up with 4*FN CONSECUTIVE columns of its row.
  unsigned vo320[2][5], vo1280[2][5], vo256[2][5];
#pragma unroll
  for (int hf = 0; hf < 2; hf++)
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const int rho = (wave + 4 * i) * 8 + lrow, fn = rho >> 4, ii = rho & 15;
      const int n0 = hf * 160 + (ii >> 2) * 40 + fn * 4 + (ii & 3);
      vo320[hf][i] = (unsigned)(n0 * 320 + cch * 8) * 2;
      vo1280[hf][i] = (unsigned)(n0 * 1280 + cch * 8) * 2;
      const int n1 = hf * 128 + (ii >> 2) * 32 + fn * 4 + (ii & 3);
      vo256[hf][i] = rho < 128 ? (unsigned)(n1 * 320 + cch * 8) * 2 : PAD_OFF;
    }

  // ---- the weight stream ---------------------------------------------------------------------------------------
  struct WT { rsrc_t r; unsigned soff; int vset; };
  auto mk = [&](const half_t* base, unsigned soff, int vset, bool live) {
    WT t;
    t.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(base), 0, live ? 0x7ffffff0 : 0, 0x00020000);
    t.soff = soff; t.vset = vset;
    return t;
  };
  // segment 0: out_proj of the self-attention, q_proj.  segment 1: cross out_proj, 10 x (GEGLU-1 chunk, GEGLU-2 chunk), conv_out.
  auto tile_at = [&](int seg, int gi) {
    if (seg == 0) {
      if (gi < 5) return mk(p.Wso, gi * 128, 0, true);
      if (gi < 10) return mk(p.Wq, (gi - 5) * 128, 0, true);
      return mk(p.Wso, 0, 0, false);
    }
    if (gi < 5) return mk(p.Wco, gi * 128, 0, true);
    if (gi < 75) {
      const int q = gi - 5, j = q / 7, r = q - 7 * j;
      if (r < 5) return mk(p.W1 + (long long)j * 256 * 320, r * 128, 1, true);
      return mk(p.W2, (unsigned)(j * 128 + (r - 5) * 64) * 2, 2, true);
    }
    if (gi < 80) return mk(p.Wout, (gi - 75) * 128, 0, true);
    return mk(p.Wso, 0, 0, false);
  };
  auto piece = [&](const WT& t, int hf, int slot, int i) {
    const unsigned v0 = hf ? vo320[1][i] : vo320[0][i], v1 = hf ? vo256[1][i] : vo256[0][i], v2 = hf ? vo1280[1][i] : vo1280[0][i];
    const unsigned v = t.vset == 0 ? v0 : (t.vset == 1 ? v1 : v2);
    blds16(t.r, v, t.soff, smem + RING_OFF + slot * HSLOT + (wave + 4 * i) * 1024);
  };
  int gt = 0, sl = 0;  // tile counter within the segment ; ring slot of the current tile's first half
  WT saved;            // descriptor of tile gt + 1
  auto seg_begin = [&](int seg) {
    gt = 0; sl = 0;
    const WT t0 = tile_at(seg, 0);
    saved = tile_at(seg, 1);
#pragma unroll
    for (int i = 0; i < 5; i++) piece(t0, 0, 0, i);
#pragma unroll
    for (int i = 0; i < 5; i++) piece(t0, 1, 1, i);
#pragma unroll
    for (int i = 0; i < 5; i++) piece(saved, 0, 2, i);
  };

  // ---- one GEMM stage: acc[2][FN] = Atile[64][nkt*64] . W^T over the next nkt tiles of the stream -------------------
  const int a_rd = (wm * 32 + rsel) * 128, w_rd = rsel * 128;
  auto gemm = [&](auto fn_c, auto seg_c, f4 (&acc)[2][10], int a_base, int nkt) {
    constexpr int FN = decltype(fn_c)::value, SEG = decltype(seg_c)::value;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < FN; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nkt; kt++) {
      wait_vm<5>();                     // both halves of tile gt have landed (one younger half-tile stays in flight)
      __builtin_amdgcn_s_barrier();     // ... for every wave; the slots of tile gt-1 are free
      asm volatile("" ::: "memory");
      const char* sA = smem + a_base + kt * A_KT;
      int ws = sl + wn; ws = ws >= 5 ? ws - 5 : ws;
      const char* sW = smem + RING_OFF + ws * HSLOT;
      h8 af[2][2], wf[2][FN];
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
        const int coff = ((kk * 4 + g) ^ key) << 4;
#pragma unroll
        for (int b = 0; b < FN; b++) wf[kk][b] = *(const h8*)(sW + w_rd + b * 2048 + coff);
#pragma unroll
        for (int a = 0; a < 2; a++) af[kk][a] = *(const h8*)(sA + a_rd + a * 2048 + coff);
      }
      __builtin_amdgcn_sched_barrier(0);
      const WT d1 = saved, d2 = tile_at(SEG, gt + 2);
      int s3 = sl + 3; s3 = s3 >= 5 ? s3 - 5 : s3;
      int s4 = sl + 4; s4 = s4 >= 5 ? s4 - 5 : s4;
      constexpr int NM = 4 * FN, GAP = NM / 11;
#pragma unroll
      for (int q = 0; q < NM; q++) {
        const int kk = q / (2 * FN), a = (q / FN) & 1, b = q % FN;
        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][b], af[kk][a], acc[a][b], 0, 0, 0);
        if ((q + 1) % GAP == 0 && (q + 1) / GAP - 1 < 10) {
          const int i = (q + 1) / GAP - 1;
          __builtin_amdgcn_sched_barrier(0);
          if (i < 5) piece(d1, 1, s3, i); else piece(d2, 0, s4, i - 5);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      saved = d2;
      gt++;
      sl = sl + 2 >= 5 ? sl - 3 : sl + 2;
    }
  };
  using I8 = std::integral_constant<int, 8>;
  using I10 = std::integral_constant<int, 10>;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- accumulator-layout helpers: lane (rsel, g) of fragment row a holds columns cbase + b*4 + r ------------------
  const int cbase = wn * 160 + g * 40;
  // add a per-column fp32 vector (bias)
  auto add_cols = [&](f4 (&v)[2][10], const float* vec) {
#pragma unroll
    for (int b = 0; b < 10; b++) {
      const f4 bv = *(const f4*)(vec + cbase + b * 4);
#pragma unroll
      for (int a = 0; a < 2; a++) v[a][b] += bv;
    }
  };
  // fp16 rows (residual sources): 40 consecutive columns = five 16-B loads per fragment row
  auto load_rows = [&](const half_t* src, int ld, f4 (&v)[2][10], bool add) {
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const int m = min(m0 + wm * 32 + a * 16 + rsel, p.M - 1);
      const half_t* rp = src + (long long)m * ld + cbase;
#pragma unroll
      for (int q = 0; q < 5; q++) {
        const h8 hv = *(const h8*)(rp + q * 8);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const float f = (float)hv[j];
          if (add) v[a][2 * q + (j >> 2)][j & 3] += f; else v[a][2 * q + (j >> 2)][j & 3] = f;
        }
      }
    }
  };
  // write fp16(f(v)) into the A tile (swizzled A-operand layout): global 16-B chunk index = wn*20 + g*5 + q
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
  // (helpers/utils.mojo:2052-2061 via :1845-1885; App.A D8).  Two-pass, exact.
  float* scr = (float*)(smem + SCR_OFF);
  auto layernorm_to_a = [&](const f4 (&v)[2][10]) {
    float s[2], mean[2], rs[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < 10; b++) t += (v[a][b][0] + v[a][b][1]) + (v[a][b][2] + v[a][b][3]);
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      s[a] = t;
      if (g == 0) scr[(wm * 32 + a * 16 + rsel) * 2 + wn] = t;
    }
    lds_barrier();  // also: every wave is past its last read of the old A tile
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const f2 pr = *(const f2*)(scr + (wm * 32 + a * 16 + rsel) * 2);
      mean[a] = (pr[0] + pr[1]) * (1.f / C);
      float t = 0.f;
#pragma unroll
      for (int b = 0; b < 10; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const float d = v[a][b][r] - mean[a]; t += d * d; }
      t += __shfl_xor(t, 16);
      t += __shfl_xor(t, 32);
      if (g == 0) scr[128 + (wm * 32 + a * 16 + rsel) * 2 + wn] = t;
    }
    lds_barrier();
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const f2 pr = *(const f2*)(scr + 128 + (wm * 32 + a * 16 + rsel) * 2);
      rs[a] = 1.f / (sqrtf((pr[0] + pr[1]) * (1.f / C)) + p.eps);
    }
    store_a_tile(v, rs[0], mean[0], rs[1], mean[1]);
  };

  // =================================================================================================================
  f4 T[2][10];    // residual stream (fp32)
  f4 acc[2][10];  // stage accumulators
  // ---- prologue: residual 1 -> T ; ao tile -> A tile ; first three half-tiles of the weight stream --------------------
  load_rows(p.tok, p.ld_tok, T, false);
  {
    const rsrc_t ra = make_rsrc(p.ao);
#pragma unroll
    for (int i = 0; i < 10; i++) {  // 40 pieces of [8 rows][128 B]: piece j = k-tile j/8, rows (j%8)*8 ..
      const int j = wave + 4 * i, kt = j >> 3, row = (j & 7) * 8 + lrow;
      const int m = min(m0 + row, p.M - 1);
      blds16(ra, (unsigned)(m * p.ld_ao + cch * 8) * 2, kt * 128, smem + A_OFF + kt * A_KT + (j & 7) * 1024);
    }
  }
  seg_begin(0);
  // ---- tok2 = ao . Wso^T + b + tok ------------------------------------------------------------------------------------
  gemm(I10{}, S0{}, acc, A_OFF, 5);
  add_cols(acc, p.bso);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) T[a][b] += acc[a][b];
  layernorm_to_a(T);
  // ---- q = LN(tok2) . Wq^T  (scaled by softmax scale * log2 e) -----------------------------------------------------------
  gemm(I10{}, S0{}, acc, A_OFF, 5);
  wait_vm<0>();   // the dead tail of segment 0 has landed: the ring region is free for the context keys
  lds_barrier();  // every wave is done reading LN(tok2)
  store_a_tile(acc, p.qscale, 0.f, p.qscale, 0.f);

  // ---- cross attention over the sample's T context keys (helpers/attention.mojo:105-115) ------------------------------
  // K tile: 5 k-tiles x [96 key rows][128 B] in the weight-operand layout.  LDS row b*16 + ii holds key
  // 32*(b>>1) + 8*(ii>>2) + 4*(b&1) + (ii&3): output lane (g, r) of the key fragments 2kk and 2kk+1 then holds keys
  // 32kk + 8g + 0..7 - exactly the A fragment of P.V k-step kk.
  {
    const rsrc_t rk = make_rsrc(p.Kc + bsmp * p.sK);
#pragma unroll
    for (int i = 0; i < 15; i++) {  // 60 pieces: k-tile j/12, rows (j%12)*8 ..
      const int j = wave + 4 * i, kt = j / 12, rho = (j - kt * 12) * 8 + lrow;
      const int b = rho >> 4, ii = rho & 15;
      const int kidx = 32 * (b >> 1) + 8 * (ii >> 2) + 4 * (b & 1) + (ii & 3);
      const unsigned v = kidx < p.T ? (unsigned)(kidx * p.ldk + cch * 8) * 2 : PAD_OFF;
      blds16(rk, v, kt * 128, smem + RING_OFF + kt * 12288 + (j - kt * 12) * 1024);
    }
  }
  wait_vm<0>();
  lds_barrier();  // q tile and K tile visible
  // per wave: rows wm*32.., heads 4*wn .. 4*wn+3 ; P (fp16 A fragments) and 1/rowsum kept for all four heads
  h8 pf[4][2][3];
  float rinv[4][2];
#pragma unroll
  for (int hh = 0; hh < 4; hh++) {
    const int h = wn * 4 + hh, c0 = h * 5;  // first 16-B chunk of the head's 40 columns
    f4 sc[2][6];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 6; b++) sc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < 2; st++) {
      // k-step 0: chunks c0 .. c0+3 ; k-step 1: chunk c0+4 in lane group 0, zeros in the key operand elsewhere
      const int cg = st == 0 ? c0 + g : c0 + 4;
      const int ktile = cg >> 3, cpos = cg & 7;
      h8 qa[2], kb[6];
#pragma unroll
      for (int a = 0; a < 2; a++) qa[a] = *(const h8*)(smem + A_OFF + ktile * A_KT + a_rd + a * 2048 + ((cpos ^ key) << 4));
#pragma unroll
      for (int b = 0; b < 6; b++) {
        kb[b] = *(const h8*)(smem + RING_OFF + ktile * 12288 + (b * 16 + rsel) * 128 + ((cpos ^ key) << 4));
        if (st == 1 && g != 0) kb[b] = h8{0, 0, 0, 0, 0, 0, 0, 0};
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) sc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb[b], qa[a], sc[a][b], 0, 0, 0);
    }
    // softmax over the keys of each query row: 24 in-lane scores x 4 lane groups (softmax with max subtraction, App.A D6)
#pragma unroll
    for (int a = 0; a < 2; a++) {
      float mx = -1.0e30f;
#pragma unroll
      for (int b = 0; b < 6; b++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kidx = 32 * (b >> 1) + 8 * g + 4 * (b & 1) + r;
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
          const half_t ph = (half_t)__builtin_amdgcn_exp2f(sc[a][2 * kk + (j >> 2)][j & 3] - mx);
          o[j] = ph;
          sum += (float)ph;  // the normaliser sums the SAME rounded probabilities the MFMA multiplies
        }
        pf[hh][a][kk] = o;
      }
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      rinv[hh][a] = 1.f / sum;
    }
  }
  lds_barrier();  // every wave is done with the K tile (and with q)
  // V^T tile: 2 k-tiles x [320 channel rows][128 B] (keys 0..63 | 64..127; chunks past the T keys are zero)
  {
    const rsrc_t rv = make_rsrc(p.Vt + bsmp * p.sVt);
    const int nch = (p.T + 7) >> 3;
#pragma unroll
    for (int i = 0; i < 20; i++) {  // 80 pieces: k-tile j/40, rows (j%40)*8 ..
      const int j = wave + 4 * i, kt = j / 40, row = (j - kt * 40) * 8 + lrow;
      const int ch = kt * 8 + cch;
      const unsigned v = ch < nch ? (unsigned)(row * p.ldvt + ch * 8) * 2 : PAD_OFF;
      blds16(rv, v, 0, smem + RING_OFF + kt * 40960 + (j - kt * 40) * 1024);
    }
  }
  wait_vm<0>();
  lds_barrier();
#pragma unroll
  for (int hh = 0; hh < 4; hh++) {
    const int h = wn * 4 + hh;
    f4 oc[2][3];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) oc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 3; kk++) {
      h8 vb[3];
#pragma unroll
      for (int b = 0; b < 3; b++) {  // channel rows h*40 + b*16 + rsel (rows past the head's 40 produce discarded outputs)
        const int row = h * 40 + b * 16 + rsel;
        vb[b] = *(const h8*)(smem + RING_OFF + (kk >> 1) * 40960 + row * 128 + ((((kk & 1) * 4 + g) ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) oc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vb[b], pf[hh][a][kk], oc[a][b], 0, 0, 0);
    }
    // attention output (merged heads) -> A tile: lane (g, r) of fragment b holds channel h*40 + b*16 + 4g + r
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
          for (int r = 0; r < 4; r++) o[r] = (half_t)(oc[a][b][r] * rinv[hh][a]);
          *(h4*)(smem + A_OFF + (col >> 6) * A_KT + row * 128 + ((((col >> 3) & 7) ^ key) << 4) + (col & 7) * 2) = o;
        }
      }
    }
  }
  lds_barrier();  // attention output complete in the A tile ; the ring region is free again
  seg_begin(1);
  // ---- tok3 = attn . Wco^T + b + tok2 -------------------------------------------------------------------------------
  gemm(I10{}, S1{}, acc, A_OFF, 5);
  add_cols(acc, p.bco);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) T[a][b] += acc[a][b];
  layernorm_to_a(T);
  // ---- GEGLU feed-forward: ten chunks of 128 hidden units; the second GEMM accumulates over the chunks ------------------
  f4 acc2[2][10];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) acc2[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  for (int jc = 0; jc < 10; jc++) {
    gemm(I8{}, S1{}, acc, A_OFF, 5);  // (a, g) interleaved: 256 columns
    {
      const float* bv = p.b1 + jc * 256 + wn * 128 + g * 32;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const int row = wm * 32 + a * 16 + rsel;
#pragma unroll
        for (int q = 0; q < 2; q++) {  // 16 activations per lane and fragment row = two 16-B chunks
          h8 o;
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const int b = q * 4 + (j >> 1), r = (j & 1) * 2;
            const f4 bb = *(const f4*)(bv + b * 4);
            o[j] = (half_t)((acc[a][b][r] + bb[r]) * gelu_tanh_c(acc[a][b][r + 1] + bb[r + 1]));
          }
          *(h8*)(smem + ACT_OFF + wn * A_KT + row * 128 + (((g * 2 + q) ^ key) << 4)) = o;
        }
      }
    }
    // the tile barrier that opens the next GEMM makes the activations visible (and 5 barriers separate this chunk's
    // reads from the next chunk's writes)
    f4 t2[2][10];
    gemm(I10{}, S1{}, t2, ACT_OFF, 2);
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 10; b++) acc2[a][b] += t2[a][b];
  }
  add_cols(acc2, p.b2);
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) T[a][b] += acc2[a][b];
  // tok4 -> A tile (every wave passed the last GEGLU-1 tile long ago: the A tile is free)
  store_a_tile(T, 1.f, 0.f, 1.f, 0.f);
  // ---- out = tok4 . Wout^T + b + x --------------------------------------------------------------------------------
  gemm(I10{}, S1{}, acc, A_OFF, 5);
  add_cols(acc, p.bout);
  load_rows(p.x, p.ld_x, acc, true);
  float gs1[4] = {0.f, 0.f, 0.f, 0.f}, gs2[4] = {0.f, 0.f, 0.f, 0.f};  // this lane's 4 groups of 10 channels
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const int m = m0 + wm * 32 + a * 16 + rsel;
    half_t* op = p.out + (long long)m * p.ld_out + cbase;
#pragma unroll
    for (int q = 0; q < 5; q++) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        o[j] = (half_t)acc[a][2 * q + (j >> 2)][j & 3];
        const float f = (float)o[j];
        const int grp = (q * 8 + j) / 10;
        gs1[grp] += f; gs2[grp] += f * f;
      }
      if (m < p.M) *(h8*)(op + q * 8) = o;
    }
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
  wait_vm<0>();  // the dead tail DMAs have landed before the workgroup's LDS is released
}

// ---- host side ----------------------------------------------------------------------------------------------------
bool attn_tail_supported(int C_, int d, int heads, int T, int64_t M, int S) {
  static const int on = getenv("TSD_CHAIN") ? atoi(getenv("TSD_CHAIN")) : 1;
  return on && C_ == 320 && d == 40 && heads == 8 && T >= 1 && T <= 96 && S % 64 == 0 && M % 64 == 0 && M < (1 << 24);
}

int launch_attn_tail(tsd_ctx* ctx, const AttnTailArgs& a) {
  if (!attn_tail_supported(a.C, a.d, a.heads, a.T, a.M, a.S)) TSD_FAIL(TSD_E_SHAPE, "attention tail: unsupported shape");
  if (a.ldw_so != 320 || a.ldw_q != 320 || a.ldw_co != 320 || a.ldw_1 != 320 || a.ldw_2 != 1280 || a.ldw_out != 320)
    TSD_FAIL(TSD_E_SHAPE, "attention tail: unexpected weight pitches");
  if (a.ld_ao % 8 || a.ld_tok % 8 || a.ld_x % 8 || a.ld_out % 8 || a.ldk % 8 || a.ldvt % 8)
    TSD_FAIL(TSD_E_SHAPE, "attention tail: misaligned pitches");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_CHAIN, (int)a.M, a.C, 0, 1);
  TailK k;
  k.ao = a.ao; k.tok = a.tok; k.x = a.x; k.out = a.out;
  k.ld_ao = a.ld_ao; k.ld_tok = a.ld_tok; k.ld_x = a.ld_x; k.ld_out = a.ld_out;
  k.Wso = a.Wso; k.Wq = a.Wq; k.Wco = a.Wco; k.W1 = a.W1; k.W2 = a.W2; k.Wout = a.Wout;
  k.bso = a.bso; k.bco = a.bco; k.b1 = a.b1; k.b2 = a.b2; k.bout = a.bout;
  k.Kc = a.Kc; k.ldk = a.ldk; k.sK = a.sK; k.Vt = a.Vt; k.ldvt = a.ldvt; k.sVt = a.sVt;
  k.T = a.T; k.M = (int)a.M; k.S = a.S;
  k.qscale = a.scale * 1.4426950408889634f; k.eps = a.eps;
  k.gn_part = a.gn_part; k.gn_nslab = a.gn_nslab;
  static bool attr = false;
  if (!attr) {
    HIP_TRY(hipFuncSetAttribute((const void*)attn_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr = true;
  }
  hipLaunchKernelGGL(attn_tail_kernel, dim3((unsigned)(a.M / BM)), dim3(256), LDS_BYTES, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
