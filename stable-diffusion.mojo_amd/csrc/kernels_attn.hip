// kernels_attn.hip - fused (flash-style) multi-head attention core for gfx950, fp16 in / fp32 softmax.
//
// Replaces the attention core of `Self_Attention.forward` (helpers/attention.mojo:30-62) and
// `Cross_Attention.forward` (:105-115): softmax(q k^T / sqrt(d_h)) v per head with the standard
// head split/merge (App.A D5) and softmax over keys with max subtraction (App.A D6; the
// reference materialises the (H,Tq,Tk) score matrix, helpers/attention.mojo:46 - here it never
// leaves registers).
//
// Design (CDNA4, wave64):
//   * one wave owns QB x 32 query rows (QB = 2 at d = 40 for long key loops, else 1); a 4-wave workgroup shares each
//     64-key K / V^T tile through LDS (descriptor LDS-DMA, double-buffered, one barrier per tile).
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_f16 ("swapped QK^T"): every lane then holds 32 of
//     the 64 scores of ONE query row, so the row max/sum are in-lane plus a single lane^32
//     exchange, and the O rescale is lane-local.
//   * softmax without per-score arithmetic: scale*log2(e) is folded into the Q fragments and -ref is the C operand of
//     the QK^T MFMAs, so the accumulator already holds the exponent; an optimistic pass keeps ref fixed after key tile 0
//     and takes no maximum (exp2 + fp16 convert per score), an exact pass repeats a workgroup only if an fp16 P
//     overflowed (see the comment at `run`).
//   * O^T = V^T . P^T: P is consumed straight from registers as the MFMA B operand.  The K rows
//     are loaded in a bit-2/bit-3-swapped order so that the 8 probabilities a lane owns per
//     k-step are 8 CONSECUTIVE keys -> the V^T operand is one ds_read_b128.
//   * V is consumed transposed ([channel][token], produced that way by the projection GEMM with
//     swapped operands) so both operands are K-contiguous; no transposing LDS reads needed.
//   * head dims that are not a multiple of 16 (d=40) are zero-padded per 16-B chunk by the DMA's range check /
//     a padded offset; LDS row pitches are odd multiples of 16 B (K) / XOR-swizzled (V^T)
//     so all ds_read_b128 are bank-conflict-free.
//   * the 4096 x 4096 d = 40 call: matrix pipe 55 % busy at 1.97-2.0 GHz (GRBM_GUI_ACTIVE / SQ_BUSY_CYCLES, profiles/r03_pmc_clock.txt -
//     the round-2 "power-bound at 1.2-1.5 GHz" reading is withdrawn); the ablation builds below (TSD_ATTN_ABL) show MFMA time, fragment
//     reads, DMA issue and plain VALU work of a SIMD's waves adding up, with the exponentials and the per-tile barrier free.
#include <stdlib.h>

#include <vector>
#include <algorithm>
#include <type_traits>
#include <atomic>

#include "attn_common.h"

// Ablation builds of the attention core (timing only, results wrong): bit 0 no exponentials, bit 1 no P.V MFMAs, bit 2 no
// QK^T MFMAs, bit 3 s_setprio(1) around the QK^T MFMAs, bit 4 s_setprio(1) around the P.V / exponential region, bit 5 no per-tile
// barrier, bit 6 no per-tile DMA wait and no barrier, bit 7 no tile DMA at all
#ifndef TSD_ATTN_ABL
#define TSD_ATTN_ABL 0
#endif

__device__ __forceinline__ void glds16a(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

#ifdef TSD_ATTN_TS
__device__ unsigned long long g_attn_ts[4 * 65536];  // per block: memtime start/end, memrealtime start/end
#endif
// QB = 32-query blocks per wave.  Every K / V^T fragment read from LDS feeds QB MFMAs, and the LDS pipe is what bounds
// QB = 1: one 1-KiB ds_read_b128 per 32x32x16 MFMA is 8 LDS cycles per 32 matrix-pipe cycles on each of 4 SIMDs - the
// CU's whole LDS bandwidth, before the tile DMA writes.  QB = 2 (d = 40: 64 queries per wave, 256 per workgroup) halves
// that at the price of 2 waves per SIMD instead of 4.

template <int D, int QB>
__global__ __launch_bounds__(256, D == 40 ? 6 - 2 * QB : 2) void flash_attn_kernel(const AttnK p) {
  constexpr int NST = 2;                   // K / V^T tile buffers
  constexpr int DCH = D / 8;               // 16-B chunks per K row
  constexpr int KSTEPS = (DCH + 1) / 2;    // QK^T k-steps (16 wide)
  // LDS pitch of a K row in 16-B chunks: odd -> conflict-free ds_read_b128.  d=40: the 5 data chunks already are an odd
  // pitch; its 6th (zero-padded) k-chunk then reads the next row's first chunk, which is harmless because the matching
  // Q chunk is zero and every byte it can touch is finite tile data (the V^T tile follows the K tile).
  constexpr int KPITCH = (DCH & 1) ? DCH : ((2 * KSTEPS) | 1);
  constexpr int DBLK = (D + 31) / 32;      // 32-wide d blocks of the output
  constexpr int VROWS = DBLK * 32;
  constexpr int K_BYTES = 64 * KPITCH * 16, V_BYTES = VROWS * 128;
  // 1-KiB DMA wave-instructions per tile.  Only rows < D of the V^T tile are streamed: the pad rows (zeros, and the
  // ones row) never change and are written once in the prologue.
  static_assert(D % 8 == 0, "head dim");
  constexpr int K_INSTR = KPITCH, V_INSTR = D / 8;
  constexpr int K_PW = (K_INSTR + 3) / 4, V_PW = (V_INSTR + 3) / 4;
  constexpr int BUF_BYTES = K_BYTES + V_BYTES;
  constexpr int FLAG_OFF = NST * BUF_BYTES;  // one word behind the tile buffers: "some row of this workgroup overflowed" (early abort, below)
  // Row sums for free: when the 32-row d blocks have a spare row (d = 40, 80) V^T row D is all ones, so the P.V MFMA
  // also accumulates l = sum_k P[q][k] in O^T row D - rescaled with O, and summed over the SAME fp16-rounded P.
  constexpr bool ONES_ROW = VROWS > D;
  constexpr int L_BLK = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3);  // accumulator holding row D (lanes hi=0)
  static_assert(!ONES_ROW || (((D % 32) >> 2) & 1) == 0, "row D must live in the hi=0 half");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef TSD_ATTN_TS
  const int ts_blk = blockIdx.x + blockIdx.y * gridDim.x;
  if (threadIdx.x == 0 && ts_blk < 65536) { g_attn_ts[ts_blk * 4] = __builtin_amdgcn_s_memtime(); g_attn_ts[ts_blk * 4 + 2] = __builtin_amdgcn_s_memrealtime(); }
#endif

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bx = blockIdx.x, bh = blockIdx.y;
  if (p.xcd_map) {  // bijective on the grid: id -> (xcd = id % 8, j = id / 8) -> query tile j % nq of head xcd + 8 * (j / nq)
    const int nq = gridDim.x, id = bx + nq * bh, j = id >> 3;
    bx = j % nq;
    bh = (id & 7) + 8 * (j / nq);
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = bx * (128 * QB) + wave * (32 * QB);

  const half_t* Qb = p.Q + b * p.sQ + h * D;
  const half_t* Kb = p.K + b * p.sK + h * D;
  const half_t* Vb = p.Vt + b * p.sVt + (long long)h * D * p.ldvt;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q][chunk ks*2+hi] -----------------
  h8 qf[QB][KSTEPS];
#pragma unroll
  for (int qb = 0; qb < QB; qb++) {
    int qrow = q0 + qb * 32 + l31;
    if (qrow >= p.Sq) qrow = p.Sq - 1;
    const half_t* qp = Qb + (long long)qrow * p.ldq;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
      const int ch = ks * 2 + hi;
      if (ch < DCH) qf[qb][ks] = *(const h8*)(qp + ch * 8);
      else qf[qb][ks] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }
  // scale * log2(e) goes into Q once (fp32 multiply, one fp16 rounding), so the MFMA result is already the exponent
  // and the per-score v_fma_f32 of the softmax disappears.  Applied AFTER tile 0's DMA has been issued (below): the Q round trip
  // and the first K / V^T round trip then overlap instead of following each other (every launch; the 77-key and 256-key
  // launches of the lower levels are a handful of such round trips long).
  auto scale_q = [&]() {
#pragma unroll
    for (int qb = 0; qb < QB; qb++)
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
        for (int e = 0; e < 8; e++) qf[qb][ks][e] = (half_t)((float)qf[qb][ks][e] * p.c);
  };

  // ---- per-thread DMA slots -------------------------------------------------------------------
  // Tiles go global -> LDS through buffer descriptors (lds_dma.h): a 32-bit byte offset per DMA instruction, the
  // tile position in the scalar offset, zero fill from the range check - no address VALU in the loop.
  // K tile: slot = row*KPITCH + pos ; LDS row `row` holds key k0 + pi(row), pi swaps bits 2 and 3.  Keys >= Sk fall
  // outside the descriptor (it ends after row Sk-1) and read as zeros; their scores are masked below anyway.
  const rsrc_t rK = make_rsrc(Kb, p.Sk * p.ldk * 2);
  unsigned k_voff[K_PW];
#pragma unroll
  for (int i = 0; i < K_PW; i++) {
    const int slot = (wave + 4 * i) * 64 + lane;
    const int row = slot / KPITCH, pos = slot - row * KPITCH;
    const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
    k_voff[i] = pos < DCH ? (unsigned)(key * p.ldk + pos * 8) * 2 : PAD_OFF;
  }
  // V^T tile: LDS row R (= channel within head) x 8 chunks; phys pos holds logical chunk pos ^ ((R>>1)&7).  A row
  // is a run of keys inside a longer pitch, so a chunk past the last valid key is NOT out of the descriptor's
  // linear range: the last tile uses its own offset set with those chunks padded.
  const rsrc_t rV = make_rsrc(Vb);
  const int ntiles = (p.Sk + 63) >> 6;
  unsigned v_voff[V_PW], v_voff_last[V_PW];
#pragma unroll
  for (int i = 0; i < V_PW; i++) {
    const int R = (wave + 4 * i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    v_voff[i] = R < D ? (unsigned)(R * p.ldvt + c * 8) * 2 : PAD_OFF;
    v_voff_last[i] = ((ntiles - 1) * 64 + c * 8 + 8 <= p.Skv) ? v_voff[i] : PAD_OFF;
  }

  auto stage = [&](int t, int buf) {
#if TSD_ATTN_ABL & 128
    if (t > 0) return;
#endif
    char* sK = smem + buf * BUF_BYTES;
    char* sV = sK + K_BYTES;
    const int k0 = t * 64;
    const bool last = t == ntiles - 1;  // wave-uniform
#pragma unroll
    for (int i = 0; i < K_PW; i++) {
      const int j = wave + 4 * i;
      if (K_INSTR % 4 == 0 || j < K_INSTR) blds16(rK, k_voff[i], (unsigned)(k0 * p.ldk) * 2, sK + j * 1024);
    }
#pragma unroll
    for (int i = 0; i < V_PW; i++) {
      const int j = wave + 4 * i;
      if (V_INSTR % 4 == 0 || j < V_INSTR) blds16(rV, last ? v_voff_last[i] : v_voff[i], (unsigned)k0 * 2, sV + j * 1024);
    }
  };

  // constant part of both V^T buffers: rows D..VROWS (zero; row D = ones when it carries the row sums)
  constexpr int PADCH = (VROWS - D) * 8 > 0 ? (VROWS - D) * 8 : 1;  // 16-B chunks of constant rows per buffer (0 at d = 160)
  for (int i = tid; i < NST * (VROWS - D) * 8; i += 256) {
    const int buf = i / PADCH, rem = i - buf * PADCH;
    const int R = D + (rem >> 3), pos = rem & 7;
    const half_t fill = (ONES_ROW && R == D) ? (half_t)1.f : (half_t)0.f;
    *(h8*)(smem + buf * BUF_BYTES + K_BYTES + R * 128 + pos * 16) = h8{fill, fill, fill, fill, fill, fill, fill, fill};
  }

  // tile 0 of the optimistic pass goes out now (the pad rows above and the DMA'd rows are disjoint), then Q is scaled
  stage(0, 0);
  scale_q();

  // ---- second optimistic reference (round 3): the query's OWN 32-key block ------------------------------------
  // Trained self-attention is peaked on a token's own neighbourhood; when that is not in key tile 0 a score can exceed tile 0's
  // row maximum by more than the 20 log2 units the optimistic pass tolerates and the whole workgroup repeats exactly (a 2x
  // cliff on this kernel).  One extra S^T block per query block (3 MFMAs, K fragments straight from global memory, once per
  // kernel) gives the row maximum over keys q0+32*qb .. +31 as a second lower bound of the true maximum.  The key set is the
  // same for the 32- and the 64-query-per-wave variants, so they stay bitwise equal.
  float mxd[QB];
#pragma unroll
  for (int qb = 0; qb < QB; qb++) {
    mxd[qb] = -1.0e30f;
    if (p.diag) {
      int krow = q0 + qb * 32 + l31;
      if (krow >= p.Sk) krow = p.Sk - 1;
      const half_t* kp = Kb + (long long)krow * p.ldk;
      f16v sd;
#pragma unroll
      for (int r = 0; r < 16; r++) sd[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const int ch = ks * 2 + hi;
        h8 kf = h8{0, 0, 0, 0, 0, 0, 0, 0};
        if (ch < DCH) kf = *(const h8*)(kp + ch * 8);
        sd = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[qb][ks], sd, 0, 0, 0);
      }
      float m = sd[0];
#pragma unroll
      for (int r = 1; r < 16; r++) m = fmaxf(m, sd[r]);
      mxd[qb] = fmaxf(m, __shfl_xor(m, 32));
    }
  }

  // Online softmax with a LAZY reference: the exponent s - ref comes straight out of the QK^T MFMAs (ref enters as their
  // C operand, a 16-register block holding -ref).  Tile 0 sets ref from its exact row maximum (so the sum cannot
  // underflow).  Two passes share the loop below:
  //  * OPTIMISTIC (always run): ref never moves after tile 0 and no maximum is taken - a tile's VALU work is exp2 and
  //    the fp16 convert only.  A later score more than 16 log2 units above ref makes its fp16 P infinite, which
  //    poisons that row's sum (the ones row of V^T / the O accumulators): checked once, after the last tile.
  //  * EXACT (only if some row of the workgroup overflowed): the whole workgroup runs again with the row maximum
  //    taken per tile and ref moved whenever a tile exceeds it by TSD_ATTN_LAZY (P <= 2^LAZY, never overflows).
  f16v o[QB][DBLK];
  float m_run[QB], l_run[QB];
  f16v nm[QB];
  const int vkey = (lane >> 1) & 7;  // swizzle key of V^T row (db*32 + l31)

  // Early abort of the optimistic pass (round 5).  An overflow used to be noticed after the LAST tile only, so a workgroup that had to
  // repeat paid a whole optimistic pass first: with peaked logits (every workgroup repeating) the kernel took 2x.  The row sum (row D of
  // O^T, fed by the ones row of V^T) is infinite from the overflowing tile on; every eighth tile each wave looks at it - one compare per
  // query block - raises a flag in LDS in front of the tile's barrier, and all waves leave together behind it.  Same decision as
  // the final check (an inf never leaves the sum), so the results are the ones of the unabridged pass; d = 160 (no ones row, at most
  // 4 tiles at the 16x16 level) keeps the final check only.
  constexpr int CHECK_EVERY = TSD_ATTN_CHECK_EVERY;  // (a huge value = the round-4 behaviour: final check only; for same-box A/B builds)
  if (tid == 0) *(volatile int*)(smem + FLAG_OFF) = 0;  // published by run()'s first barrier
  auto run = [&](auto exact_c) {
  constexpr bool EXACT = decltype(exact_c)::value;
#pragma unroll
  for (int qb = 0; qb < QB; qb++) {
#pragma unroll
    for (int d = 0; d < DBLK; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[qb][d][r] = 0.f;
    m_run[qb] = 0.f; l_run[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r++) nm[qb][r] = 0.f;
  }
  if (EXACT) stage(0, 0);  // the optimistic pass' tile 0 was issued before Q was scaled
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; t++) {
    tsd_jitter();
    f16v s[QB][2];
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
    {  // S^T - ref = K . Q^T + (-ref) of one 64-key tile: two 32-key blocks per query block, K fragments read once
      const char* sK = smem + (t & 1) * BUF_BYTES;
#if TSD_ATTN_ABL & 8
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const h8 k0f = *(const h8*)(sK + ((l31)*KPITCH + ks * 2 + hi) * 16);
        const h8 k1f = *(const h8*)(sK + ((32 + l31) * KPITCH + ks * 2 + hi) * 16);
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
#if TSD_ATTN_ABL & 4
          asm volatile("" ::"v"(k0f), "v"(k1f));
          s[qb][0] = nm[qb]; s[qb][1] = nm[qb];
#else
          s[qb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k0f, qf[qb][ks], ks == 0 ? nm[qb] : s[qb][0], 0, 0, 0);
          s[qb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(k1f, qf[qb][ks], ks == 0 ? nm[qb] : s[qb][1], 0, 0, 0);
#endif
        }
      }
#if TSD_ATTN_ABL & 8
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    // a use after the first-k-step MFMAs: keeps them in the untied (dst != C) form, no copy of nm
#pragma unroll
    for (int qb = 0; qb < QB; qb++) keep_alive(nm[qb]);
    const char* sV = smem + (t & 1) * BUF_BYTES + K_BYTES;
    // lane (hi, r) of block kb holds key kb*32 + (r&3) + 4*((r>>2)&1) + 8*hi + 16*(r>>3)
    if ((t + 1) * 64 > p.Sk) {
      const int k0 = t * 64;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int kl = kb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
          if (k0 + kl >= p.Sk) {
#pragma unroll
            for (int qb = 0; qb < QB; qb++) s[qb][kb][r] = -1.0e30f;
          }
        }
    }
    if (EXACT || t == 0) {
      float mx[QB];
      bool move = t == 0;
#pragma unroll
      for (int qb = 0; qb < QB; qb++) {
        float m = s[qb][0][0];
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
          for (int r = 0; r < 16; r++) m = fmaxf(m, s[qb][kb][r]);
        mx[qb] = fmaxf(m, __shfl_xor(m, 32));
        move = move || mx[qb] > TSD_ATTN_LAZY;
      }
      if (__any(move)) {  // wave-uniform: move the reference (always on tile 0, then rarely)
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
          const float delta = t == 0 ? (EXACT ? mx[qb] : fmaxf(mx[qb], mxd[qb]) + TSD_ATTN_HEADROOM) : fmaxf(mx[qb], 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] += delta;
#pragma unroll
          for (int r = 0; r < 16; r++) nm[qb][r] = -m_run[qb];
#pragma unroll
          for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) s[qb][kb][r] -= delta;
          if (t != 0) {
            if (!ONES_ROW) l_run[qb] *= alpha;
            const f2 a2 = {alpha, alpha};
#pragma unroll
            for (int d = 0; d < DBLK; d++)
#pragma unroll
              for (int r = 0; r < 16; r += 2) {
                f2 v = {o[qb][d][r], o[qb][d][r + 1]};
                v *= a2;
                o[qb][d][r] = v[0]; o[qb][d][r + 1] = v[1];
              }
          }
        }
      }
    }
    f2 psum2[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) psum2[qb] = f2{0.f, 0.f};
    // P^T chunk kq (8 keys per lane) feeds the P.V MFMAs of chunk kq only, so the exponentials of chunk kq+1 are issued
    // between those MFMAs: the matrix pipe works through chunk kq while the VALU (exp2 is the slow part) produces the
    // next one.  The issue order is pinned; left to itself the compiler emits all the exponentials and then all MFMAs.
    auto p_part = [&](int qb, int kq, int half, h8& pf) {  // 4 of the 8 probabilities of chunk kq
      const int kb = kq >> 1, r0 = (kq & 1) * 8 + half * 4;
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
#if TSD_ATTN_ABL & 1
        const float p0 = s[qb][kb][r0 + r] * 0.001f, p1 = s[qb][kb][r0 + r + 1] * 0.001f;
#else
        const float p0 = __builtin_amdgcn_exp2f(s[qb][kb][r0 + r]), p1 = __builtin_amdgcn_exp2f(s[qb][kb][r0 + r + 1]);
#endif
        if (!ONES_ROW) psum2[qb] += f2{p0, p1};
        pf[half * 4 + r] = (half_t)p0;
        pf[half * 4 + r + 1] = (half_t)p1;
      }
    };
    auto v_frag = [&](int kq, h8 (&vf)[DBLK]) {
#pragma unroll
      for (int d = 0; d < DBLK; d++) vf[d] = *(const h8*)(sV + (d * 32 + l31) * 128 + (((kq * 2 + hi) ^ vkey) << 4));
    };
    h8 vf_cur[DBLK], vf_next[DBLK];
#if TSD_ATTN_ABL & 16
    __builtin_amdgcn_s_setprio(1);
#endif
    v_frag(0, vf_cur);
    h8 pf_cur[QB], pf_next[QB];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) { p_part(qb, 0, 0, pf_cur[qb]); p_part(qb, 0, 1, pf_cur[qb]); }
#pragma unroll
    for (int kq = 0; kq < 4; kq++) {  // kq = kb*2 + half ; logical chunk = kq*2 + hi
      if (kq < 3) v_frag(kq + 1, vf_next);  // LDS latency hides under this chunk's MFMAs
      __builtin_amdgcn_sched_barrier(0);
      // QB * DBLK MFMAs; the 2 * QB half-chunks of exponentials for chunk kq+1 are spread between them
      constexpr int NM = QB * DBLK, NP = 2 * QB;
#pragma unroll
      for (int i = 0; i < NM; i++) {
        const int qb = i / DBLK, d = i - qb * DBLK;
#if TSD_ATTN_ABL & 2
        asm volatile("" ::"v"(vf_cur[d]), "v"(pf_cur[qb]));
#else
        o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf_cur[d], pf_cur[qb], o[qb][d], 0, 0, 0);
#endif
        if (kq < 3) {
#pragma unroll
          for (int j = (i * NP) / NM; j < ((i + 1) * NP) / NM; j++) p_part(j >> 1, kq + 1, j & 1, pf_next[j >> 1]);
        }
        if (QB > 1) __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kq < 3) {
#pragma unroll
        for (int qb = 0; qb < QB; qb++) pf_cur[qb] = pf_next[qb];
#pragma unroll
        for (int d = 0; d < DBLK; d++) vf_cur[d] = vf_next[d];
      }
    }
#if TSD_ATTN_ABL & 16
    __builtin_amdgcn_s_setprio(0);
#endif
    if (!ONES_ROW) {
#pragma unroll
      for (int qb = 0; qb < QB; qb++) l_run[qb] += psum2[qb][0] + psum2[qb][1];
    }
#if TSD_ATTN_ABL & 32   // ablation: no per-tile barrier (each wave only waits for its own DMA pieces)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif TSD_ATTN_ABL & 64 // ablation: neither the DMA wait nor the barrier
#else
    // (the EXACT pass never consults the flag - `check` is false there - so the word needs no reset between the passes: ADVICE r05)
    const bool check = !EXACT && ONES_ROW && (t % CHECK_EVERY) == CHECK_EVERY - 1 && t + 1 < ntiles;  // uniform
    if constexpr (!EXACT && ONES_ROW) {
      if (check) {
        bool over = false;
#pragma unroll
        for (int qb = 0; qb < QB; qb++) over = over || !(o[qb][L_BLK][L_REG] < 1.0e37f);  // inf / NaN (the hi = 1 lanes hold a pad row: 0, or NaN next to an inf P)
        if (__any(over) && lane == 0) *(volatile int*)(smem + FLAG_OFF) = 1;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (check && *(volatile int*)(smem + FLAG_OFF) != 0) break;  // every wave reads the same word behind the same barrier
#endif
  }
  };
  // row sum l: row D of O^T when V^T carries the ones row (it sits in the hi=0 lane of each query's lane pair)
  auto row_sum = [&](int qb) {
    if constexpr (ONES_ROW) {
      const float lv = o[qb][L_BLK][L_REG];
      const float lo = __shfl_xor(lv, 32);
      return hi ? lo : lv;
    }
    return l_run[qb] + __shfl_xor(l_run[qb], 32);
  };
  run(std::false_type{});
  float l_tot[QB];
  {
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
      l_tot[qb] = row_sum(qb);
      bad = bad || !(l_tot[qb] < 1.0e37f);  // inf or NaN
      if (!ONES_ROW) {  // the fp32 sum of fp32 P cannot see an fp16 P that overflowed; O can
        float amax = 0.f, osum = 0.f;
#pragma unroll
        for (int d = 0; d < DBLK; d++)
#pragma unroll
          for (int r = 0; r < 16; r++) {
            amax = fmaxf(amax, fabsf(o[qb][d][r]));  // NaN-dropping max finds inf; 0 * x finds NaN
            osum += o[qb][d][r] * 0.f;
          }
        bad = bad || !(amax < 1.0e37f) || osum != 0.f;
      }
    }
    if (__syncthreads_or(bad)) {
      if (tid == 0) atomicAdd(p.exact_ctr, 1);
      run(std::true_type{});
#pragma unroll
      for (int qb = 0; qb < QB; qb++) l_tot[qb] = row_sum(qb);
    }
  }
#ifdef TSD_ATTN_TS
  if (threadIdx.x == 0 && ts_blk < 65536) { g_attn_ts[ts_blk * 4 + 1] = __builtin_amdgcn_s_memtime(); g_attn_ts[ts_blk * 4 + 3] = __builtin_amdgcn_s_memrealtime(); }
#endif

  // ---- normalise and store O[b][q][h*D + d] -----------------------------------------------------
#pragma unroll
  for (int qb = 0; qb < QB; qb++) {
    const float inv = 1.f / l_tot[qb];
    const int qrow = q0 + qb * 32 + l31;
    if (qrow < p.Sq) {
      half_t* op = p.O + b * p.sO + (long long)qrow * p.ldo + h * D;
#pragma unroll
      for (int d = 0; d < DBLK; d++)
#pragma unroll
        for (int g4 = 0; g4 < 4; g4++) {
          const int dbase = d * 32 + 8 * g4 + 4 * hi;
          if (dbase < D) {
            h4 v;
#pragma unroll
            for (int r = 0; r < 4; r++) v[r] = (half_t)(o[qb][d][g4 * 4 + r] * inv);
            *(h4*)(op + dbase) = v;
          }
        }
    }
  }
}

bool attn_fused_supported(int d) { return d == 40 || d == 80 || d == 160; }
// per-context switches (TsdOptions): attn_diag 0 = optimistic reference from key tile 0 only (the round-2 behaviour);
// attn_qb_force 0 = kernel chosen by the layer shape, 1 / 2 = that many query blocks per wave (4-wave workgroups) whenever d = 40,
// 3 = the 8-wave two-group kernel whenever d = 40
extern "C" int tsd_debug_set_attn_diag(tsd_ctx* ctx, int on) {
  return ctx_set_option(ctx, &TsdOptions::attn_diag, on, 0, 1);
}
extern "C" int tsd_debug_set_attn_qb(tsd_ctx* ctx, int mode) {
  return ctx_set_option(ctx, &TsdOptions::attn_qb_force, mode, 0, 3);
}
template <int D, int QB>
static int launch_fa(tsd_ctx* ctx, const AttnK& k, int B, int H, int Sq) {
  constexpr int DCH = D / 8, KSTEPS = (DCH + 1) / 2, KPITCH = (DCH & 1) ? DCH : ((2 * KSTEPS) | 1), DBLK = (D + 31) / 32;
  constexpr int LDS = 2 * (64 * KPITCH * 16 + DBLK * 32 * 128) + 16;  // + the early-abort flag word
  auto fn = flash_attn_kernel<D, QB>;
  static std::atomic<unsigned long long> attr{0};  // one bit per device
  if (!((attr.load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(fn, dim3(ceil_div(Sq, 128 * QB), B * H), dim3(256), LDS, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_flash_attention(tsd_ctx* ctx, const AttnArgs& a) {
  if (!attn_fused_supported(a.d)) TSD_FAIL(TSD_E_SHAPE, "flash attention: head dim %d unsupported", a.d);
  if (a.ldq % 8 || a.ldk % 8 || a.ldvt % 8 || a.ldo % 4) TSD_FAIL(TSD_E_SHAPE, "flash attention: misaligned pitches");
  if (a.Sq <= 0 || a.Sk <= 0) TSD_FAIL(TSD_E_SHAPE, "flash attention: empty sequence");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ATTN, a.Sq, a.Sk, a.d, a.B * a.H);
  AttnK k;
  k.Q = a.Q; k.K = a.K; k.Vt = a.Vt; k.O = a.O; k.zeros = ctx->zeros; k.ones = ctx->zeros + 1024;
  k.sQ = a.sQ; k.sK = a.sK; k.sVt = a.sVt; k.sO = a.sO;
  k.ldq = a.ldq; k.ldk = a.ldk; k.ldvt = a.ldvt; k.ldo = a.ldo;
  k.H = a.H; k.Sq = a.Sq; k.Sk = a.Sk; k.Skv = std::min(round_up(a.Sk, 8), a.ldvt);
  k.c = a.scale * 1.4426950408889634f;
  k.exact_ctr = ctx->status + 2;
  k.diag = (ctx->opt.attn_diag && a.Sq == a.Sk) ? 1 : 0;
  // long key loops only (the K / V^T stream of a head is what the remap saves); results do not depend on the block order.
  // Measured (profiles/r03_attn_xcd_ab.txt): FETCH_SIZE of the 4096 x 4096 d = 40 call 366 -> 144 MB, of the 1024 x 1024 d = 80 call
  // 103 -> 33 MB (-0.9 GB of a step's 10.9 GB) and 201.1 against 201.5 steps/s - the kernel is not bound by that stream (all of it
  // Infinity-Cache hits) and eight whole heads per XCD (5.2 MB of K / V^T) no longer fit its 4 MB L2.  OFF by default (TSD_ATTN_XCD=1).
  k.xcd_map = (ctx->opt.attn_xcd && (a.B * a.H) % 8 == 0 && a.Sk >= 512) ? 1 : 0;
  switch (a.d) {
    case 40:
      // 64 queries per wave when the key loop is long enough to matter: 4096 x 4096 at B*H = 64 runs 253 -> 243 us (half the
      // K / V^T fragment reads per MFMA); 77-key cross attention is better off with the 128-query workgroup
      // The choice keys on the layer (Sq, Sk, H), never on the batch: the two variants are bitwise equal only while no
      // workgroup repeats exactly (the repeat is decided per 128- / 256-query workgroup and moves the reference per 32 / 64
      // rows), so a sample computed alone must run the same variant as its row of a batch (bitwise batch invariance).
      // Eight waves in two staggered groups (kernels_attn8.hip) for the long key loops of a big grid - the 64x64 level's 4096 x 4096 call
      if (ctx->opt.attn_qb_force ? ctx->opt.attn_qb_force == 3 : (ctx->opt.attn_wg8 && a.Sk >= 512 && (long long)ceil_div(a.Sq, 512) * a.H >= 64))
        return launch_flash_attention8(ctx, k, a.B, a.H, a.Sq, a.d, ctx->opt.attn8_var);
      if (ctx->opt.attn_qb_force ? ctx->opt.attn_qb_force == 2 : (ctx->opt.attn_qb == 2 && a.Sk >= 512 && (long long)ceil_div(a.Sq, 256) * a.H >= 64))
        return launch_fa<40, 2>(ctx, k, a.B, a.H, a.Sq);
      return launch_fa<40, 1>(ctx, k, a.B, a.H, a.Sq);
    case 80: return launch_fa<80, 1>(ctx, k, a.B, a.H, a.Sq);
    default: return launch_fa<160, 1>(ctx, k, a.B, a.H, a.Sq);
  }
}

// Workgroups of flash_attn_kernel that ran the exact pass on this context since the last reset (synchronises the stream).
extern "C" int tsd_debug_attn_exact_passes(tsd_ctx* ctx, int reset) {
  if (!ctx || !ctx->status) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
  int v = 0;
  if (hipMemcpy(&v, ctx->status + 2, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  if (reset) { const int z = 0; if (hipMemcpy(ctx->status + 2, &z, sizeof(z), hipMemcpyHostToDevice) != hipSuccess) return -1; }
  return v < 0 ? 0x7fffffff : v;
}

// Debug/bench entry: time `iters` launches of the fused attention core on synthetic device data.
extern "C" int tsd_debug_attn_bench(tsd_ctx* ctx, int B, int H, int d, int Sq, int Sk, int iters, float* ms) {
  if (!ctx || !ms || iters <= 0) TSD_FAIL(TSD_E_ARG, "attn_bench: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const int C = H * d, Skp = round_up(Sk, 8);
  const int64_t nq = (int64_t)B * Sq * C, nk = (int64_t)B * Sk * C, nv = (int64_t)B * C * Skp;
  TSD_TRY(ctx_reserve_arena(ctx, (size_t)(2 * nq + nk + nv) * 2 + (size_t)std::max(nq, std::max(nk, nv)) * 4 + 8192));
  ctx->arena.top = 0;
  half_t* Q = arena_alloc<half_t>(ctx, nq);
  half_t* K = arena_alloc<half_t>(ctx, nk);
  half_t* Vt = arena_alloc<half_t>(ctx, nv);
  half_t* O = arena_alloc<half_t>(ctx, nq);
  float* tmp = arena_alloc<float>(ctx, std::max(nq, std::max(nk, nv)));
  if (!Q || !K || !Vt || !O || !tmp) TSD_FAIL(TSD_E_ALLOC, "attn_bench: arena");
  TSD_TRY(launch_fill_uniform(ctx, tmp, nq, 1, 11, 2.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)nq, Q, (int)nq, 1));
  TSD_TRY(launch_fill_uniform(ctx, tmp, nk, 1, 12, 2.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)nk, K, (int)nk, 1));
  TSD_TRY(launch_fill_uniform(ctx, tmp, nv, 1, 13, 1.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)nv, Vt, (int)nv, 1));
  AttnArgs a;
  a.Q = Q; a.ldq = C; a.sQ = (int64_t)Sq * C; a.K = K; a.ldk = C; a.sK = (int64_t)Sk * C;
  a.Vt = Vt; a.ldvt = Skp; a.sVt = (int64_t)C * Skp; a.O = O; a.ldo = C; a.sO = (int64_t)Sq * C;
  a.B = B; a.H = H; a.d = d; a.Sq = Sq; a.Sk = Sk; a.scale = 1.f / sqrtf((float)d);
  int r = launch_flash_attention(ctx, a);
  if (r != TSD_OK) return r;
  HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < iters && r == TSD_OK; i++) r = launch_flash_attention(ctx, a);
  HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
  HIP_TRY(hipEventSynchronize(ctx->ev1));
  float t = 0.f;
  HIP_TRY(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
  *ms = t / iters;
#ifdef TSD_ATTN_TS
  {
    const int nblk = std::min(65536, ((Sq + 127) / 128) * B * H);
    std::vector<unsigned long long> h((size_t)nblk * 4);
    HIP_TRY(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_attn_ts), h.size() * 8));
    double clk = 0, dur = 0; int n = 0;
    unsigned long long r0 = ~0ull, r1 = 0;
    for (int b = 0; b < nblk; b++) {
      const double dt = (double)(h[b * 4 + 1] - h[b * 4]), dr = (double)(h[b * 4 + 3] - h[b * 4 + 2]);
      if (dr > 0) { clk += dt / (dr * 10.0); dur += dr * 10.0; n++; }  // realtime ticks are 10 ns
      r0 = std::min(r0, h[b * 4 + 2]); r1 = std::max(r1, h[b * 4 + 3]);
    }
    fprintf(stderr, "[attn ts] blocks=%d mean block loop %.1f us, s_memtime ticks per ns %.3f, first-start..last-end %.1f us\n", n,
            dur / n * 1e-3, clk / n, (r1 - r0) * 0.01);
  }
#endif
  ctx->arena.top = 0;
  return r;
}
