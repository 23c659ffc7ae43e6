// kernels_attn.hip - fused (flash-style) multi-head attention core for gfx950, fp16 in / fp32 softmax.
//
// Replaces the attention core of `Self_Attention.forward` (helpers/attention.mojo:30-62) and
// `Cross_Attention.forward` (:105-115): softmax(q k^T / sqrt(d_h)) v per head with the standard
// head split/merge (App.A D5) and softmax over keys with max subtraction (App.A D6; the
// reference materialises the (H,Tq,Tk) score matrix, helpers/attention.mojo:46 - here it never
// leaves registers).
//
// Design (CDNA4, wave64):
//   * one wave owns 32 query rows; a 4-wave workgroup shares each 64-key K / V^T tile through LDS
//     (global_load_lds_dwordx4 DMA, double-buffered, one barrier per tile).
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_f16 ("swapped QK^T"): every lane then holds 32 of
//     the 64 scores of ONE query row, so the row max/sum are in-lane plus a single lane^32
//     exchange, and the O rescale is lane-local.
//   * O^T = V^T . P^T: P is consumed straight from registers as the MFMA B operand.  The K rows
//     are loaded in a bit-2/bit-3-swapped order so that the 8 probabilities a lane owns per
//     k-step are 8 CONSECUTIVE keys -> the V^T operand is one ds_read_b128.
//   * V is consumed transposed ([channel][token], produced that way by the projection GEMM with
//     swapped operands) so both operands are K-contiguous; no transposing LDS reads needed.
//   * head dims that are not a multiple of 16 (d=40) are zero-padded per 16-B chunk by pointing the
//     DMA source at a zero page; LDS row pitches are odd multiples of 16 B (K) / XOR-swizzled (V^T)
//     so all ds_read_b128 are bank-conflict-free.
#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

struct AttnK {
  const half_t* Q; const half_t* K; const half_t* Vt; half_t* O; const half_t* zeros;
  long long sQ, sK, sVt, sO;
  int ldq, ldk, ldvt, ldo;
  int H, Sq, Sk, Skv;  // Skv: number of valid V^T columns (Sk rounded up to 8)
  float c;             // scale * log2(e)
};

__device__ __forceinline__ void glds16a(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int D>
// min 2 waves/SIMD: caps the budget at 256 unified registers so the MFMA results stay in VGPRs (no v_accvgpr moves
// around the softmax / rescale VALU work).
__global__ __launch_bounds__(256, 2) void flash_attn_kernel(const AttnK p) {
  constexpr int DCH = D / 8;               // 16-B chunks per K row
  constexpr int KSTEPS = (DCH + 1) / 2;    // QK^T k-steps (16 wide)
  constexpr int KPITCH = (2 * KSTEPS) | 1; // LDS pitch of a K row in chunks (odd -> conflict-free)
  constexpr int DBLK = (D + 31) / 32;      // 32-wide d blocks of the output
  constexpr int VROWS = DBLK * 32;
  constexpr int K_BYTES = 64 * KPITCH * 16, V_BYTES = VROWS * 128;
  constexpr int K_INSTR = KPITCH, V_INSTR = VROWS / 8;  // 1-KiB wave instructions per tile
  constexpr int K_PW = (K_INSTR + 3) / 4, V_PW = (V_INSTR + 3) / 4;
  constexpr int BUF_BYTES = K_BYTES + V_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int bh = blockIdx.y, b = bh / p.H, h = bh - b * p.H;
  const int q0 = blockIdx.x * 128 + wave * 32;

  const half_t* Qb = p.Q + b * p.sQ + h * D;
  const half_t* Kb = p.K + b * p.sK + h * D;
  const half_t* Vb = p.Vt + b * p.sVt + (long long)h * D * p.ldvt;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane holds Q[q][chunk ks*2+hi] -----------------
  h8 qf[KSTEPS];
  {
    int qrow = q0 + l31;
    if (qrow >= p.Sq) qrow = p.Sq - 1;
    const half_t* qp = Qb + (long long)qrow * p.ldq;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks++) {
      const int ch = ks * 2 + hi;
      if (ch < DCH) qf[ks] = *(const h8*)(qp + ch * 8);
      else qf[ks] = h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
  }

  // ---- per-thread DMA slots -------------------------------------------------------------------
  // K tile: slot = row*KPITCH + pos ; LDS row `row` holds key k0 + pi(row), pi swaps bits 2 and 3.
  int k_key[K_PW]; int k_off[K_PW]; bool k_data[K_PW];
#pragma unroll
  for (int i = 0; i < K_PW; i++) {
    const int slot = (wave + 4 * i) * 64 + lane;
    const int row = slot / KPITCH, pos = slot - row * KPITCH;
    const int key = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
    k_key[i] = key;
    k_off[i] = key * p.ldk + pos * 8;
    k_data[i] = pos < DCH;
  }
  // V^T tile: LDS row R (= channel within head) x 8 chunks; phys pos holds logical chunk pos ^ ((R>>1)&7)
  int v_chunk[V_PW]; long long v_off[V_PW]; bool v_data[V_PW];
#pragma unroll
  for (int i = 0; i < V_PW; i++) {
    const int R = (wave + 4 * i) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    v_chunk[i] = c;
    v_off[i] = (long long)R * p.ldvt + c * 8;
    v_data[i] = R < D;
  }
  const half_t* zsrc = p.zeros;

  auto stage = [&](int t, int buf) {
    char* sK = smem + buf * BUF_BYTES;
    char* sV = sK + K_BYTES;
    const int k0 = t * 64;
#pragma unroll
    for (int i = 0; i < K_PW; i++) {
      const int j = wave + 4 * i;
      if (K_INSTR % 4 == 0 || j < K_INSTR) {
        const bool ok = k_data[i] && (k0 + k_key[i] < p.Sk);
        const half_t* src = ok ? Kb + (long long)k0 * p.ldk + k_off[i] : zsrc;
        glds16a(src, sK + j * 1024);
      }
    }
#pragma unroll
    for (int i = 0; i < V_PW; i++) {
      const int j = wave + 4 * i;
      if (V_INSTR % 4 == 0 || j < V_INSTR) {
        const bool ok = v_data[i] && (k0 + v_chunk[i] * 8 + 8 <= p.Skv);
        const half_t* src = ok ? Vb + k0 + v_off[i] : zsrc;
        glds16a(src, sV + j * 1024);
      }
    }
  };

  f16v o[DBLK];
#pragma unroll
  for (int d = 0; d < DBLK; d++)
#pragma unroll
    for (int r = 0; r < 16; r++) o[d][r] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  const int vkey = (lane >> 1) & 7;  // swizzle key of V^T row (db*32 + l31)
  const int ntiles = (p.Sk + 63) >> 6;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < ntiles; t++) {
    if (t + 1 < ntiles) stage(t + 1, (t + 1) & 1);
    const char* sK = smem + (t & 1) * BUF_BYTES;
    const char* sV = sK + K_BYTES;

    // ---- S^T = K . Q^T : two 32-key blocks ------------------------------------------------------
    f16v s[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
#pragma unroll
      for (int r = 0; r < 16; r++) s[kb][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ks++) {
        const h8 kf = *(const h8*)(sK + ((kb * 32 + l31) * KPITCH + ks * 2 + hi) * 16);
        s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[kb], 0, 0, 0);
      }
    }
    // lane (hi, r) of block kb holds key kb*32 + (r&3) + 4*((r>>2)&1) + 8*hi + 16*(r>>3)
    if ((t + 1) * 64 > p.Sk) {
      const int k0 = t * 64;
#pragma unroll
      for (int kb = 0; kb < 2; kb++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          const int kl = kb * 32 + (r & 3) + 4 * ((r >> 2) & 1) + 8 * hi + 16 * (r >> 3);
          if (k0 + kl >= p.Sk) s[kb][r] = -1.0e30f;
        }
    }
    // ---- online softmax (fp32) ----------------------------------------------------------------
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * p.c);
    const float mc = m_new * p.c;
    m_run = m_new;
    float psum = 0.f;
    h8 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; kb++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const float pv = __builtin_amdgcn_exp2f(s[kb][r] * p.c - mc);
        psum += pv;
        pf[kb * 2 + (r >> 3)][r & 7] = (half_t)pv;
      }
    l_run = l_run * alpha + psum;
#pragma unroll
    for (int d = 0; d < DBLK; d++)
#pragma unroll
      for (int r = 0; r < 16; r++) o[d][r] *= alpha;
    // ---- O^T += V^T . P^T ------------------------------------------------------------------------
#pragma unroll
    for (int d = 0; d < DBLK; d++) {
#pragma unroll
      for (int kq = 0; kq < 4; kq++) {  // kq = kb*2 + half ; logical chunk = kq*2 + hi
        const h8 vf = *(const h8*)(sV + (d * 32 + l31) * 128 + (((kq * 2 + hi) ^ vkey) << 4));
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[kq], o[d], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- normalise and store O[b][q][h*D + d] -----------------------------------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.f / l_tot;
  const int qrow = q0 + l31;
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
          for (int r = 0; r < 4; r++) v[r] = (half_t)(o[d][g4 * 4 + r] * inv);
          *(h4*)(op + dbase) = v;
        }
      }
  }
}

bool attn_fused_supported(int d) { return d == 40 || d == 80 || d == 160; }

template <int D>
static int launch_fa(tsd_ctx* ctx, const AttnK& k, int B, int H, int Sq) {
  constexpr int DCH = D / 8, KSTEPS = (DCH + 1) / 2, KPITCH = (2 * KSTEPS) | 1, DBLK = (D + 31) / 32;
  constexpr int LDS = 2 * (64 * KPITCH * 16 + DBLK * 32 * 128);
  auto fn = flash_attn_kernel<D>;
  static bool attr = false;
  if (!attr) {
    HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr = true;
  }
  hipLaunchKernelGGL(fn, dim3(ceil_div(Sq, 128), B * H), dim3(256), LDS, ctx->stream, k);
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
  k.Q = a.Q; k.K = a.K; k.Vt = a.Vt; k.O = a.O; k.zeros = ctx->zeros;
  k.sQ = a.sQ; k.sK = a.sK; k.sVt = a.sVt; k.sO = a.sO;
  k.ldq = a.ldq; k.ldk = a.ldk; k.ldvt = a.ldvt; k.ldo = a.ldo;
  k.H = a.H; k.Sq = a.Sq; k.Sk = a.Sk; k.Skv = std::min(round_up(a.Sk, 8), a.ldvt);
  k.c = a.scale * 1.4426950408889634f;
  switch (a.d) {
    case 40: return launch_fa<40>(ctx, k, a.B, a.H, a.Sq);
    case 80: return launch_fa<80>(ctx, k, a.B, a.H, a.Sq);
    default: return launch_fa<160>(ctx, k, a.B, a.H, a.Sq);
  }
}
