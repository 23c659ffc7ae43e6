// kernels_attn8.hip - flash attention for the long d = 40 key loops (the 64x64 level's 4096 x 4096 self-attention: the largest single
// kernel of a denoise step) as an EIGHT-wave workgroup whose two halves run one phase apart.
//
// Same arithmetic as flash_attn_kernel<40, 2> (kernels_attn.hip; `Self_Attention.forward`, helpers/attention.mojo:30-62): swapped QK^T on
// v_mfma_f32_32x32x16_f16 with scale*log2(e) folded into Q, optimistic softmax pass with an exact repeat, P consumed from registers as the B
// operand of O^T = V^T.P^T, row sums from a ones row of V^T; every accumulator sees the same products in the same order.  One arithmetic
// difference: d = 40 pads the QK^T reduction to 48, and this kernel uses the first pad column for the softmax reference - K column 40 is the
// constant 32 (a 16-byte block per ring slot that the pad chunk's lanes read), Q column 40 holds -ref/32 - so the scores leave the MFMAs as
// s - ref WITHOUT a 16-register C operand per query block (32 VGPRs the schedule below needs).  The reference is thereby rounded to 32 x fp16
// (any reference gives the same softmax; m_run holds exactly what was subtracted), so the results agree with flash_attn_kernel to the fp16
// rounding of P, not bit for bit.
//
// What the schedule is built on (scripts/micro/valu_port.hip, profiles/r06_valu_port*.txt - the two waves of a SIMD, shader clocks):
//   * MFMAs and VALU work do not overlap by themselves.  28 MFMAs take 960 clocks, 64 v_exp_f32 + 32 v_cvt_pk_f16_f32 620; one wave doing both
//     takes the sum (1260 with half of the VALU work), two waves doing both side by side take 2390 for twice the work: a wave whose next
//     instruction is an MFMA waiting for the matrix pipe keeps the SIMD's VALU issue.  That - not LDS reads, DMA or the barrier - is why
//     flash_attn_kernel's matrix pipe is 52-55 % busy (profiles/r05_pmc_sq.txt): its two waves per SIMD serialise.
//   * A partner's VALU work does hide behind a wave's MFMAs when that wave drops to priority 0 for one instruction behind every MFMA
//     (s_setprio 0 ; s_setprio 1): 28 MFMAs + the partner's 64 exp + 32 cvt in 1190 clocks, whichever wave is older; the flip costs the MFMA
//     chain ~10 clocks each.  A wave's OWN VALU work between its MFMAs always adds.
// So a wave alternates between
//   * an X phase (no MFMA): this wave's LDS-DMA pieces of key tile t+2, the eight V^T fragment reads of tile t, the exponentials + converts
//     that turn the first three chunks of query block 0's scores S(t) into P(t) (24 of the tile's 64 exponentials), and
//   * an M phase (28 MFMAs at priority 1): per query block P(t).V(t) (8) then K(t+1).Q^T (6); query block 0's last chunk rides behind the first
//     four MFMAs, query block 1's 32 exponentials between MFMAs 4..17, and behind each of the last ten - which carry no VALU work of their own -
//     the wave drops to priority 0 for one instruction (the split and the flip placement found by measurement, profiles/r06_attn8_*: all 64
//     exponentials in the X phase make it the long pole, flips behind every MFMA cost 50 clocks per tile more); the six K(t+1) fragment reads
//     go out behind the first MFMA,
// and waves 0-3 / 4-7, which sit pairwise on the four SIMDs, run the two phases in opposition under two s_barrier per tile.  All eight waves
// share each K / V^T tile (512 queries per workgroup: half the LDS-DMA per query of the 4-wave kernel); 3-slot ring; group 0 issues the five
// 1-KiB DMA instructions of a K tile, group 1 those of a V^T tile (one per wave, a second one on each group's first wave) behind counted waits.
// Measured (profiles/r06_attn8_*.txt): 4096 x 4096, B*H = 64: 2500 clocks per key tile and SIMD against 3230 for flash_attn_kernel<40, 2>;
// 6-9 % less time on the same box (the chip gives most of the cycle saving back as clock).
//
// Ring protocol (h = half periods; group 0: X(t) at h = 2t, M(t) at 2t+1; group 1: X(t) at 2t+1, M(t) at 2t+2):
//   K(t+2) is issued by group 0 at the start of X(t), waited for (counted: the pieces of tile t+3 stay in flight) at the end of X(t+1),
//   i.e. in front of the barrier that opens h = 2t+3, the first half period that reads it (group 0's M(t+1)); its slot held K(t-1), last
//   read by group 1 in M(t-2) at h = 2t-2.  V(t+2) is issued by group 1 at the start of its X(t) (h = 2t+1), waited for in front of the
//   barrier that opens h = 2t+4 = group 0's X(t+2), the first reader; its slot held V(t-1), last read by group 1 at h = 2t-1.
#include <stdlib.h>

#include <vector>
#include <algorithm>
#include <type_traits>
#include <atomic>

#include "attn_common.h"

// variant bits (timing / A-B builds, -DTSD_ATTN8_VARIANTS): 1 = static s_setprio(1) for waves 4-7, 2 = s_setprio(1) around every M phase,
// 4 = no exponentials (timing only), 8 = no MFMAs (timing only), 16 x n = n of query block 1's four chunks exponentiated in the X phase,
// 128 = priority flip behind every MFMA of the M phase, 256 = V^T fragment reads at the start of the X phase, 512 = flips only behind the MFMAs
// without own exponentials, 1024 = query block 0's fourth chunk exponentiated in the M phase
#ifndef TSD_ATTN8_DEFAULT_VAR
#define TSD_ATTN8_DEFAULT_VAR 1920  // shipped: 128 + 256 + 512 + 1024 (flips behind the MFMAs that carry no own VALU work, V^T fragments first, X 24 + M 40 exponentials)
#endif
#ifdef TSD_ATTN8_TS
__device__ unsigned long long g_attn8_ts[256 * 8 * 4];  // per (block < 256, wave): ticks in X, at barrier 1, in M, at barrier 2
#endif

template <int D, int VAR>
__global__ __launch_bounds__(512, 2) void flash_attn8_kernel(const AttnK p) {
  static_assert(D == 40, "the 8-wave kernel is built for d = 40 (5 + 5 DMA pieces per tile split evenly over 4 + 4 waves)");
  constexpr int QB = 2, NB = 3;
  constexpr int XCH = (VAR >> 4) & 7;  // chunks (of 4) of query block 1 exponentiated in the X phase; the others between the M phase's MFMAs
  static_assert(XCH <= 4, "variant");
  constexpr bool VF_EARLY = (VAR & 256) != 0;  // V^T fragment reads at the start of the X phase instead of behind the exponentials
  constexpr bool Q0LATE = (VAR & 1024) != 0;  // query block 0's fourth chunk exponentiated in the M phase (X 24 + M 40 instead of 32 + 32)
  constexpr bool FLIP = (VAR & 128) != 0;  // priority 1 in the M phase, dropped for one instruction behind every MFMA
  constexpr int DCH = D / 8, KSTEPS = (DCH + 1) / 2, KPITCH = (DCH & 1) ? DCH : ((2 * KSTEPS) | 1);
  constexpr int DBLK = (D + 31) / 32, VROWS = DBLK * 32;
  constexpr int K_BYTES = 64 * KPITCH * 16, V_BYTES = VROWS * 128, KPAD_OFF = K_BYTES + V_BYTES, BUF_BYTES = KPAD_OFF + 16;  // slot: K tile, V^T tile, the K pad chunk
  constexpr int FLAG_OFF = NB * BUF_BYTES;
  constexpr int L_BLK = D / 32, L_REG = ((D % 32) & 3) + 4 * ((D % 32) >> 3);  // accumulator holding row D = the row sums (lanes hi = 0)
  static_assert(VROWS > D && (((D % 32) >> 2) & 1) == 0, "ones row in the hi = 0 half");
  // 1-KiB DMA instructions per tile: 64 rows x KPITCH chunks (K), D rows x 8 chunks (V^T) = five each; piece j goes to wave j of the
  // tile's group, the fifth piece to its wave 0 as well (counted waits: two alternatives, lds_dma.h)
  constexpr int K_INSTR = KPITCH, V_INSTR = D / 8;
  static_assert(K_INSTR == 5 && V_INSTR == 5, "five pieces per tile and group: waves 0..3 one each, wave 0 a second one");
  constexpr int CHECK_EVERY = TSD_ATTN_CHECK_EVERY;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, gw = wave & 3;  // waves w and w + 4 share a SIMD (a workgroup's waves are dealt to the SIMDs cyclically)
  const int hi = lane >> 5, l31 = lane & 31;
  int bx = blockIdx.x, bh = blockIdx.y;
  if (p.xcd_map) {
    const int nq = gridDim.x, id = bx + nq * bh, j = id >> 3;
    bx = j % nq;
    bh = (id & 7) + 8 * (j / nq);
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int q0 = bx * 512 + wave * 64;

  const half_t* Qb = p.Q + b * p.sQ + h * D;
  const half_t* Kb = p.K + b * p.sK + h * D;
  const half_t* Vb = p.Vt + b * p.sVt + (long long)h * D * p.ldvt;
  const int ntiles = (p.Sk + 63) >> 6;

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

  // ---- this wave's one or two DMA instructions per tile: group 0 the K tile, group 1 the V^T tile -------------------
  // K tile: slot = row*KPITCH + pos; LDS row `row` holds key k0 + pi(row), pi swaps bits 2 and 3 (a lane's 8 probabilities per k-step are
  // then 8 consecutive keys); keys >= Sk fall outside the descriptor and read as zeros.  V^T tile: slot = R*8 + pos, pos holds logical
  // chunk pos ^ ((R>>1)&7); chunks past the last valid key of the last tile are padded by their own offset set.
  const rsrc_t rT = make_rsrc(grp ? (const void*)Vb : (const void*)Kb, grp ? 0x7ffffff0 : p.Sk * p.ldk * 2);
  const unsigned tile_stride = grp ? 128u : 128u * (unsigned)p.ldk;
  const bool two = gw == 0;  // this wave issues two pieces per tile (wave-uniform)
  unsigned voff[2], voff_last[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int slot = (i == 0 ? gw : 4) * 64 + lane;
    if (grp == 0) {
      const int row = slot / KPITCH, pos = slot - row * KPITCH;
      const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
      voff[i] = (unsigned)(key * p.ldk + pos * 8) * 2;
      voff_last[i] = voff[i];
    } else {
      const int R = slot >> 3;
      const int c = (slot & 7) ^ ((R >> 1) & 7);
      voff[i] = (unsigned)(R * p.ldvt + c * 8) * 2;
      voff_last[i] = ((ntiles - 1) * 64 + c * 8 + 8 <= p.Skv) ? voff[i] : PAD_OFF;
    }
  }
  auto stage = [&](int t, int buf) {
    char* dst = smem + buf * BUF_BYTES + (grp ? K_BYTES : 0);
    const unsigned soff = (unsigned)t * tile_stride;
    const bool last = t == ntiles - 1;  // wave-uniform
    blds16(rT, last ? voff_last[0] : voff[0], soff, dst + gw * 1024);
    if (two) blds16(rT, last ? voff_last[1] : voff[1], soff, dst + 4096);
  };

  // constant part of the V^T buffers: rows D..VROWS (zero; row D = ones: it carries the row sums)
  for (int i = tid; i < NB * (VROWS - D) * 8; i += 512) {
    const int buf = i / ((VROWS - D) * 8), rem = i - buf * ((VROWS - D) * 8);
    const int R = D + (rem >> 3), pos = rem & 7;
    const half_t fill = R == D ? (half_t)1.f : (half_t)0.f;
    *(h8*)(smem + buf * BUF_BYTES + K_BYTES + R * 128 + pos * 16) = h8{fill, fill, fill, fill, fill, fill, fill, fill};
  }
  if (tid < NB) *(h8*)(smem + tid * BUF_BYTES + KPAD_OFF) = h8{(half_t)32.f, 0, 0, 0, 0, 0, 0, 0};  // K chunk 5 of every key: column 40 = 32 (REF_UNIT), 41..47 = 0
  // "some row of this workgroup overflowed" (early abort); an LDS-typed pointer: a volatile access through the generic one becomes a flat load
  volatile __attribute__((address_space(3))) int* const flag = (volatile __attribute__((address_space(3))) int*)(smem + FLAG_OFF);
  if (tid == 0) *flag = 0;  // published by run()'s first barrier

  // tiles 0 and 1 of the optimistic pass go out now, then Q is scaled: the Q round trip and the first tile round trips overlap
  stage(0, 0);
  if (ntiles > 1) stage(1, 1);
#pragma unroll
  for (int qb = 0; qb < QB; qb++)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++)
#pragma unroll
      for (int e = 0; e < 8; e++) qf[qb][ks][e] = (half_t)((float)qf[qb][ks][e] * p.c);

  // second optimistic reference: the query's OWN 32-key block (see kernels_attn.hip)
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

  f16v o[QB][DBLK];
  float m_run[QB];  // the reference in use, exactly as the MFMAs subtract it (-ref/32 in fp16, times the K pad column's 32)
  constexpr float REF_UNIT = 32.f, REF_MAX = 2.0e6f;
  const int vkey = (lane >> 1) & 7;  // swizzle key of V^T row (db*32 + l31)
#ifdef TSD_ATTN8_TS
  unsigned long long ts_x = 0, ts_b1 = 0, ts_m = 0, ts_b2 = 0;
#define TS8(acc, t0) { const unsigned long long t1__ = __builtin_amdgcn_s_memtime(); acc += t1__ - t0; t0 = t1__; }
#else
#define TS8(acc, t0)
#endif

  // One pass over the key tiles.  OPTIMISTIC (always): the reference is fixed after tile 0 and no maximum is taken; a score more than 16
  // log2 units above it makes its fp16 P infinite, which poisons the row sum - looked at every CHECK_EVERY tiles and after the last one.
  // EXACT (only if some row of the workgroup overflowed): the row maximum is taken per tile (in the X phase) and the reference moves
  // whenever a tile exceeds it by TSD_ATTN_LAZY.
  auto run = [&](auto exact_c) {
    constexpr bool EXACT = decltype(exact_c)::value;
    f16v s[QB][2];
    h8 pf[QB][4], vf[4][DBLK];
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
#pragma unroll
      for (int d = 0; d < DBLK; d++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[qb][d][r] = 0.f;
      m_run[qb] = 0.f;
      if (hi) qf[qb][KSTEPS - 1][0] = (half_t)0.f;
    }
    if (EXACT) { stage(0, 0); if (ntiles > 1) stage(1, 1); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // the six K fragments of a tile (S^T(t) - ref = K(t) . Q^T: two 32-key blocks per query block, fragments shared by both query blocks); in the last k-step the hi = 1 lanes hold the pad chunk (columns 40..47): they read the slot's constant block
    auto k_frags = [&](int buf, h8 (&kf)[KSTEPS][2]) {
      const char* sK = smem + buf * BUF_BYTES;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const bool pad = ks * 2 + 1 >= DCH;
        kf[ks][0] = *(const h8*)(sK + ((pad && hi) ? KPAD_OFF : ((l31)*KPITCH + ks * 2 + hi) * 16));
        kf[ks][1] = *(const h8*)(sK + ((pad && hi) ? KPAD_OFF : ((32 + l31) * KPITCH + ks * 2 + hi) * 16));
      }
    };
    auto qk = [&](const h8 (&kf)[KSTEPS][2]) {
      f16v z;
#pragma unroll
      for (int r = 0; r < 16; r++) z[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
          if constexpr (VAR & 8) {
            asm volatile("" ::"v"(kf[ks][0]), "v"(kf[ks][1]), "v"(qf[qb][ks]));
            if (ks == 0) { s[qb][0] = z; s[qb][1] = z; }
          } else {
            s[qb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][0], qf[qb][ks], ks == 0 ? z : s[qb][0], 0, 0, 0);
            s[qb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][1], qf[qb][ks], ks == 0 ? z : s[qb][1], 0, 0, 0);
          }
        }
      }
    };
    // scores of keys >= Sk (last, partial tile): lane (hi, r) of block kb holds key kb*32 + (r&3) + 4*((r>>2)&1) + 8*hi + 16*(r>>3)
    auto mask_tail = [&](int t) {
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
    };
    // the reference: set from tile 0's row maximum (t == 0), in the exact pass moved whenever a later tile exceeds it by TSD_ATTN_LAZY
    auto reference = [&](int t) {
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
      if (__any(move)) {  // wave-uniform
#pragma unroll
        for (int qb = 0; qb < QB; qb++) {
          const float want = m_run[qb] + (t == 0 ? (EXACT ? mx[qb] : fmaxf(mx[qb], mxd[qb]) + TSD_ATTN_HEADROOM) : fmaxf(mx[qb], 0.f));
          // the reference the MFMAs can subtract: a multiple of 32 x fp16 (small ones snap to 0: no fp16 subnormals in the operand)
          const half_t qref = (half_t)(fminf(fmaxf(want, -REF_MAX), REF_MAX) * (1.f / REF_UNIT));
          const float ref = fabsf(want) < 0.0625f ? 0.f : (float)qref * REF_UNIT;
          const float delta = ref - m_run[qb];
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_run[qb] = ref;
          if (hi) qf[qb][KSTEPS - 1][0] = (half_t)(-ref * (1.f / REF_UNIT));
#pragma unroll
          for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int r = 0; r < 16; r++) s[qb][kb][r] -= delta;
          if (t != 0) {
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
    };

    {  // tile 0: scores and the reference, all eight waves together
      h8 kf[KSTEPS][2];
      k_frags(0, kf);
      qk(kf);
      mask_tail(0);
      reference(0);
    }
    if (grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }  // group 1 runs one phase behind

    int cur = 0;  // ring slot of tile t
#ifdef TSD_ATTN8_TS
    unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
    bool aborted = false;
    auto tile = [&](int t, auto live_c, auto more_c) {
      constexpr bool LIVE = decltype(live_c)::value;  // tile t + 2 exists
      constexpr bool MORE = decltype(more_c)::value;  // tile t + 1 exists
      const int nx1 = cur == NB - 1 ? 0 : cur + 1, nx2 = cur == 0 ? NB - 1 : cur - 1;  // slots of tiles t + 1, t + 2
      // ---------------- X(t): no MFMA.  DMA pieces of tile t + 2, S(t) -> P(t) for query block 0, V^T fragments of tile t ----------------
      tsd_jitter();
      if constexpr (LIVE) stage(t + 2, nx2);
      auto v_frags = [&]() {  // the eight V^T fragments of tile t
        const char* sV = smem + cur * BUF_BYTES + K_BYTES;
#pragma unroll
        for (int kq = 0; kq < 4; kq++)
#pragma unroll
          for (int d = 0; d < DBLK; d++) vf[kq][d] = *(const h8*)(sV + (d * 32 + l31) * 128 + (((kq * 2 + hi) ^ vkey) << 4));
      };
      if constexpr (VF_EARLY) { v_frags(); __builtin_amdgcn_sched_barrier(0); }  // in front of the exponentials: their LDS latency hides under them
      if (t > 0) {
        mask_tail(t);
        if constexpr (EXACT) reference(t);
      }
      // one pair of probabilities: two exponentials and one packed convert.  A lone wave issues a v_exp_f32 every ~16 clocks (measured: 64 of
      // them back to back made the X phase 1300+ clocks against 900 of MFMAs in the partner's M phase), so only query block 0 (+ XCH chunks
      // of block 1) is exponentiated here; the rest rides between the MFMAs of this wave's own M phase.
      auto p_pair = [&](int qb, int pr) {  // pr = 0..15: chunk kq = pr / 4, elements 2 * (pr % 4), + 1
        const int kq = pr >> 2, e = (pr & 3) * 2, kb = kq >> 1, r0 = (kq & 1) * 8;
        float p0, p1;
        if constexpr (VAR & 4) { p0 = s[qb][kb][r0 + e] * 0.001f; p1 = s[qb][kb][r0 + e + 1] * 0.001f; }
        else { p0 = __builtin_amdgcn_exp2f(s[qb][kb][r0 + e]); p1 = __builtin_amdgcn_exp2f(s[qb][kb][r0 + e + 1]); }
        pf[qb][kq][e] = (half_t)p0;
        pf[qb][kq][e + 1] = (half_t)p1;
      };
      constexpr int XP = 4 * XCH;  // pairs of query block 1 done in the X phase
      constexpr int X0 = Q0LATE ? 12 : 16;  // pairs of query block 0 done here (Q0LATE: its last chunk rides behind the first four MFMAs of the M phase)
#pragma unroll
      for (int pr = 0; pr < X0; pr++) p_pair(0, pr);
#pragma unroll
      for (int pr = 0; pr < XP; pr++) p_pair(1, pr);
      // complete in front of barrier 1: without the (empty) uses hipcc sinks exponentials into the M phase
#pragma unroll
      for (int kq = 0; kq < X0 / 4; kq++) asm volatile("" : "+v"(pf[0][kq]));
#pragma unroll
      for (int kq = 0; kq < XCH; kq++) asm volatile("" : "+v"(pf[1][kq]));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!VF_EARLY) v_frags();
      // early abort: the row sum (row D of O^T, tiles < t) is infinite from an overflowing tile on.  Every CHECK_EVERY-th tile each wave looks
      // at it and raises the flag in front of barrier 1; every wave reads it behind ITS barrier 2 of the same tile - by then both groups'
      // looks at tile t are behind a barrier, and the next look is eight tiles away: all waves take the same decision.  (The exact pass never
      // consults the flag: `check` is false there.)
      const bool check = !EXACT && (t % CHECK_EVERY) == CHECK_EVERY - 1 && t + 1 < ntiles;  // uniform
      if constexpr (!EXACT) {
        if (check) {
          bool over = false;
#pragma unroll
          for (int qb = 0; qb < QB; qb++) over = over || !(o[qb][L_BLK][L_REG] < 1.0e37f);  // inf / NaN (the hi = 1 lanes hold a pad row)
          if (__any(over) && lane == 0) *flag = 1;
        }
      }
      // tile t + 1 (this wave's pieces) has landed; the pieces of tile t + 2 issued above stay in flight
      if constexpr (LIVE) {
        wait_alt_begin();  // exactly one of the two runs
        if (two) wait_vm_counted<2, 0, 2, 1>(); else wait_vm_counted<1, 0, 1, 2>();
        wait_alt_end();
      } else {
        wait_vm_counted<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      TS8(ts_x, ts0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      TS8(ts_b1, ts0);
      // ---------------- M(t): 28 MFMAs.  Per query block: O^T += V^T(t).P^T(t) (8), then S^T(t+1) = K(t+1).Q^T - ref (6) ----------------
      // slots 0-7 P.V of block 0, 8-13 QK^T of block 0 (its scores are dead since the X phase), 14-21 P.V of block 1, 22-27 QK^T of block 1.
      // Block 1's remaining exponentials are spread over slots 0..17, chunk kq complete before its first P.V MFMA (slot 14 + 2 kq) and all
      // of them before slot 22 overwrites the scores; the six K(t+1) fragment reads go out behind the first MFMA.  The issue order is
      // pinned (one scheduling region per slot).  Every accumulator sees the same products in the same order as in flash_attn_kernel.
      tsd_jitter();
      if constexpr ((VAR & 2) || FLIP) __builtin_amdgcn_s_setprio(1);
      h8 kf[KSTEPS][2];
      f16v z;
#pragma unroll
      for (int r = 0; r < 16; r++) z[r] = 0.f;
      constexpr int EXP_SLOTS = 18;
#pragma unroll
      for (int i = 0; i < 28; i++) {
        const int qb = i < 14 ? 0 : 1, j = i < 14 ? i : i - 14;
        if (j < 8) {
          const int kq = j >> 1, d = j & 1;
          if constexpr (VAR & 8) asm volatile("" ::"v"(vf[kq][d]), "v"(pf[qb][kq]));
          else o[qb][d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[kq][d], pf[qb][kq], o[qb][d], 0, 0, 0);
        } else if constexpr (MORE) {
          const int ks = (j - 8) >> 1, kb = (j - 8) & 1;
          if constexpr (VAR & 8) {
            asm volatile("" ::"v"(kf[ks][kb]), "v"(qf[qb][ks]));
            if (ks == 0) s[qb][kb] = z;
          } else {
            s[qb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[ks][kb], qf[qb][ks], ks == 0 ? z : s[qb][kb], 0, 0, 0);
          }
        }
        // A wave whose next instruction is an MFMA waiting for the matrix pipe keeps the SIMD's VALU issue to itself: its partner's exponentials
        // and converts then run AFTER the MFMAs, not under them (scripts/micro/valu_port.hip: 28 MFMAs next to 64 exp + 32 cvt take 1410-1610
        // clocks, the sum).  Dropping to priority 0 for one instruction behind every MFMA hands the partner the issue slots of the ~28 clocks
        // the pipe is busy anyway: 1190 clocks for both, whichever wave is older (the flip itself costs the MFMA chain ~10 clocks each).
        if constexpr (FLIP) { if (!(VAR & 512) || i >= EXP_SLOTS) { __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_s_setprio(1); } }  // (512: only behind MFMAs that carry none of the wave's own exponentials)
        if constexpr (MORE) {
          if (i == 0) k_frags(nx1, kf);
        }
        if constexpr (Q0LATE) {  // query block 0's last chunk: needed by slot 6, its scores are overwritten from slot 8
          if (i < 4) p_pair(0, 12 + i);
        }
#pragma unroll
        for (int pr = XP; pr < 16; pr++)
          if ((Q0LATE ? 4 : 0) + ((pr - XP) * (EXP_SLOTS - (Q0LATE ? 4 : 0))) / (16 - XP) == i) p_pair(1, pr);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((VAR & 2) || FLIP) __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      TS8(ts_m, ts0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      TS8(ts_b2, ts0);
      cur = nx1;
      if (check && *flag != 0) aborted = true;
    };
    int t = 0;
    for (; t + 2 < ntiles && !aborted; t++) tile(t, std::true_type{}, std::true_type{});
    if (t + 1 < ntiles && !aborted) { tile(t, std::false_type{}, std::true_type{}); t++; }
    if (t < ntiles && !aborted) tile(t, std::false_type{}, std::false_type{});
    if (!grp) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }  // group 0 waits for group 1's last phase
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (after an abort: pieces of the tiles ahead)
  };

  if constexpr (VAR & 1) {
    if (grp) __builtin_amdgcn_s_setprio(1);  // the second-dispatched half loses every arbitration otherwise
  }
  // row sum l: row D of O^T (it sits in the hi = 0 lane of each query's lane pair)
  auto row_sum = [&](int qb) {
    const float lv = o[qb][L_BLK][L_REG];
    const float lo = __shfl_xor(lv, 32);
    return hi ? lo : lv;
  };
  run(std::false_type{});
  float l_tot[QB];
  {
    bool bad = false;
#pragma unroll
    for (int qb = 0; qb < QB; qb++) {
      l_tot[qb] = row_sum(qb);
      bad = bad || !(l_tot[qb] < 1.0e37f);  // inf or NaN
    }
    if (__syncthreads_or(bad)) {
      if (tid == 0) atomicAdd(p.exact_ctr, 1);
      run(std::true_type{});
#pragma unroll
      for (int qb = 0; qb < QB; qb++) l_tot[qb] = row_sum(qb);
    }
  }
  if constexpr (VAR & 1) __builtin_amdgcn_s_setprio(0);
#ifdef TSD_ATTN8_TS
  {
    const int blk = blockIdx.x + blockIdx.y * gridDim.x;
    if (lane == 0 && blk < 256) {
      unsigned long long* d = g_attn8_ts + (blk * 8 + wave) * 4;
      d[0] = ts_x; d[1] = ts_b1; d[2] = ts_m; d[3] = ts_b2;
    }
  }
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

template <int D, int VAR>
static int launch_fa8(tsd_ctx* ctx, const AttnK& k, int B, int H, int Sq) {
  constexpr int DCH = D / 8, KSTEPS = (DCH + 1) / 2, KPITCH = (DCH & 1) ? DCH : ((2 * KSTEPS) | 1), DBLK = (D + 31) / 32;
  constexpr int LDS = 3 * (64 * KPITCH * 16 + DBLK * 32 * 128 + 16) + 16;  // three slots (K tile, V^T tile, K pad chunk) + the early-abort flag word
  auto fn = flash_attn8_kernel<D, VAR>;
  static std::atomic<unsigned long long> attr{0};  // one bit per device
  if (!((attr.load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr.fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
  }
  hipLaunchKernelGGL(fn, dim3(ceil_div(Sq, 512), B * H), dim3(512), LDS, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_flash_attention8(tsd_ctx* ctx, const AttnK& k, int B, int H, int Sq, int d, int variant) {
  if (d != 40) TSD_FAIL(TSD_E_SHAPE, "8-wave flash attention: head dim %d unsupported", d);
  switch (variant) {
#ifdef TSD_ATTN8_VARIANTS  // timing / A-B builds only
    case 384: return launch_fa8<40, 384>(ctx, k, B, H, Sq);
    case 896: return launch_fa8<40, 896>(ctx, k, B, H, Sq);
    case 1920: return launch_fa8<40, 1920>(ctx, k, B, H, Sq);
    case 1408: return launch_fa8<40, 1408>(ctx, k, B, H, Sq);
#endif
    default: return launch_fa8<40, TSD_ATTN8_DEFAULT_VAR>(ctx, k, B, H, Sq);
  }
}

#ifdef TSD_ATTN8_TS
extern "C" int tsd_debug_attn8_ticks(double* out16) {  // mean ticks per wave of groups 0 / 1 in X, at barrier 1, in M, at barrier 2 over blocks < 256
  std::vector<unsigned long long> h(256 * 8 * 4);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_attn8_ts), h.size() * 8) != hipSuccess) return -1;
  for (int g = 0; g < 2; g++)
    for (int k = 0; k < 4; k++) {
      double s = 0;
      for (int b = 0; b < 256; b++)
        for (int w = 0; w < 4; w++) s += (double)h[((b * 8) + g * 4 + w) * 4 + k];
      out16[g * 4 + k] = s / (256 * 4);
    }
  return 0;
}
#endif
