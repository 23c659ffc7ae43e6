// kernels_rowln.hip - Linear (+ bias, + residual) and the LayerNorm that follows it as ONE kernel, for C = 640 token rows.
//
// In `Unet_Attention_Block.forward` (diffusion.mojo:112-147) three projections are each followed by a LayerNorm of their
// output: conv_in -> LN (:117,:122), self-attention out_proj + residual -> LN (:125-129), cross-attention out_proj +
// residual -> LN (:132-136).  At the 32x32 level (C = 640, 8192 token rows at batch 8) the unfused graph runs each pair as
// a 64x160-tile GEMM (17.5 us, of which ~5 us are MFMA) and a LayerNorm launch (6.5 us): four column tiles own one row,
// so the row statistics need a second kernel.  Here a workgroup owns 32 COMPLETE rows:
//   * 4 waves side by side, each 32 rows x 160 columns (FM = 2, FN = 10, v_mfma_f32_16x16x32_f16 with swapped operands -
//     the wave tile and register layout of kernels_chain.hip);
//   * the A rows (32 x 640 fp16, 40 KB) are gathered into LDS once; the weight matrix streams through a 3-slot ring of
//     [640 rows][32 k] tiles (40 KB each) that are byte images of their LDS layout, pre-packed once per model
//     (launch_row_ln_pack): a tile is 40 linear 1-KiB buffer_load ... lds copies, two tiles ahead of the MFMAs;
//   * the epilogue adds bias / residual in the accumulator layout, keeps the row in fp32, merges the four per-wave
//     (mean, M2) pairs through LDS (Chan et al.), and writes BOTH the projection output (the next residual) and its
//     LayerNorm ((x - mean) / (sigma + eps), population sigma, no affine: helpers/utils.mojo:2052-2061 via :1845-1885,
//     App.A D8) as whole rows staged through LDS.
// Each workgroup streams the whole 800 KB weight matrix from L2, so the kernel is bound by the per-CU fetch rate
// (~113 GB/s, scripts/micro/cu_load_rate.hip: >= 7.2 us) - which is why this only pays where a launch costs more than
// that: not at C = 1280 (3.2 MB per workgroup, 16 rows each to fill the chip), and C = 320 has kernels_chain.hip.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

#include "common.h"
#include "lds_dma.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct RowLnK {
  const half_t* A; int lda;
  const half_t* wstream;
  const float* bias;
  const half_t* R; int ldr;  // residual rows (nullptr: none)
  half_t* tok; int ld_tok;   // A . W^T + bias (+ R)
  half_t* ln; int ld_ln;     // LayerNorm of that
  float eps;
};

namespace {
constexpr int CW = 640, BM = 32, NT = CW / 32;
constexpr int A_OFF = 0, A_KT = BM * 128, A_BYTES = (CW / 64) * A_KT;  // A tile: 10 k-tiles x [32 rows][128 B], XOR-swizzled
constexpr int RING_OFF = A_BYTES, TILE = CW * 64, NSLOT = 3;           // weight ring: [4 waves][160 rows][64 B] per tile
constexpr int LDS_BYTES = RING_OFF + NSLOT * TILE;
static_assert(LDS_BYTES <= 163840, "LDS budget");
constexpr int STREAM_BYTES = NT * TILE;
constexpr int RP = 1296;  // staging row pitch: 1280 B + 16 (conflict-free 16-B writes down a column of rows)
static_assert(2 * BM * RP <= NSLOT * TILE, "staging tiles must fit in the ring region");

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
// bank swizzle of the 64-B weight rows (same as kernels_chain.hip)
__host__ __device__ __forceinline__ int wswz(int rho) { return 3 * ((rho >> 2) & 1); }
}  // namespace

// weight pre-packing: W[640][ldw] (reference Linear / 1x1-conv layout, k contiguous) -> the tile stream.  One thread per
// 16-B chunk: tile t, wave part, LDS row rho, physical chunk pc  <-  W[n][32t + 8*(pc ^ wswz(rho)) ..], with
// n = part*160 + (ii>>2)*40 + fn*4 + (ii&3), fn = rho>>4, ii = rho&15: output lane (g, r) of fragment b then owns column
// part*160 + g*40 + b*4 + r - 40 consecutive columns per lane.
__global__ void k_row_ln_pack(const half_t* W, int ldw, half_t* dst) {
  const int ci = blockIdx.x * blockDim.x + threadIdx.x;
  if (ci >= STREAM_BYTES / 16) return;
  int byte = ci * 16;
  const int t = byte / TILE; byte -= t * TILE;
  const int part = byte / 10240, rb = byte - part * 10240;
  const int rho = rb >> 6, pc = (rb >> 4) & 3, fn = rho >> 4, ii = rho & 15;
  const int n = part * 160 + (ii >> 2) * 40 + fn * 4 + (ii & 3);
  const int k = 32 * t + 8 * (pc ^ wswz(rho));
  *(h8*)(dst + (size_t)ci * 8) = *(const h8*)(W + (size_t)n * ldw + k);
}

__global__ __launch_bounds__(256, 1) void row_ln_kernel(const RowLnK p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rsel = lane & 15, key = lane & 7, g = lane >> 4;
  const int lrow = lane >> 3, cch = (lane & 7) ^ lrow;  // gather DMA: LDS row within an 8-row piece, logical 16-B chunk fetched
  const int m0 = blockIdx.x * BM;
  const int cbase = wave * 160 + g * 40;  // this lane's 40 output columns: cbase + b*4 + r

  // ---- everything this workgroup reads, in queue order: A rows, residual rows, bias, weight tiles 0 and 1 ------------
  {
    const rsrc_t ra = make_rsrc(p.A);
#pragma unroll
    for (int i = 0; i < 10; i++) {  // 40 pieces of [8 rows][128 B]: piece j = k-tile j/4, rows (j%4)*8 ..
      const int j = wave + 4 * i, kt = j >> 2, row = (j & 3) * 8 + lrow;
      blds16(ra, (unsigned)((m0 + row) * p.lda + cch * 8) * 2, kt * 128, smem + A_OFF + kt * A_KT + (j & 3) * 1024);
    }
  }
  h8 raw[2][5];
  const bool has_res = p.R != nullptr;  // uniform
  if (has_res) {
#pragma unroll
    for (int a = 0; a < 2; a++) {
      const half_t* rp = p.R + (long long)(m0 + a * 16 + rsel) * p.ldr + cbase;
#pragma unroll
      for (int q = 0; q < 5; q++) raw[a][q] = *(const h8*)(rp + q * 8);
    }
  }
  f4 bv[10];
#pragma unroll
  for (int b = 0; b < 10; b++) bv[b] = *(const f4*)(p.bias + cbase + b * 4);
  const rsrc_t rw = make_rsrc(p.wstream, STREAM_BYTES);
  const unsigned lane16 = lane * 16;
  auto stage_tile = [&](int t, int slot) {  // this wave's 10 of the tile's 40 1-KiB pieces
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int j = wave + 4 * i;
      blds16(rw, lane16, (unsigned)(t * TILE + j * 1024), smem + RING_OFF + slot * TILE + j * 1024);
    }
  };
  stage_tile(0, 0);
  stage_tile(1, 1);

  f4 acc[2][10];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int b = 0; b < 10; b++) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
  const int a_rd = rsel * 128;
  const int w_rd = wave * 10240 + rsel * 64 + ((g ^ wswz(rsel)) << 4);
  int sl = 0;
  for (int t = 0; t < NT; t++) {
    // tile t has landed once only the 10 pieces of tile t+1 (issued one step ago) are still in flight; everything older
    // (A rows, residual, bias) landed with tile 0
    if (t < NT - 1) wait_vm<10>(); else wait_vm<0>();
    lds_barrier();  // every wave's pieces of tile t are visible; every wave is done reading tile t-1
    const char* sA = smem + A_OFF + (t >> 1) * A_KT + a_rd + ((((t & 1) * 4 + g) ^ key) << 4);
    const char* sW = smem + RING_OFF + sl * TILE + w_rd;
    h8 af[2], wf[10];
#pragma unroll
    for (int a = 0; a < 2; a++) af[a] = *(const h8*)(sA + a * 2048);
#pragma unroll
    for (int b = 0; b < 10; b++) wf[b] = *(const h8*)(sW + b * 1024);
    if (t + 2 < NT) stage_tile(t + 2, sl == 0 ? 2 : sl - 1);  // (sl + 2) % 3: the slot tile t-1 just left
#pragma unroll
    for (int b = 0; b < 10; b++)
#pragma unroll
      for (int a = 0; a < 2; a++) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[b], af[a], acc[a][b], 0, 0, 0);
    sl = sl == 2 ? 0 : sl + 1;
  }

  // ---- epilogue: bias (+ residual), row statistics, two whole-row outputs --------------------------------------------
  f4 T[2][10];
#pragma unroll
  for (int a = 0; a < 2; a++)
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int b = 2 * q + (j >> 2), r = j & 3;
        T[a][b][r] = acc[a][b][r] + bv[b][r] + (has_res ? (float)raw[a][q][j] : 0.f);
      }
  lds_barrier();  // every wave is done with the A tile and the ring: they become scratch and output staging
  float* scr = (float*)(smem + A_OFF);  // [32 rows][4 waves] x (mean, M2)
  constexpr int ST0 = RING_OFF, ST1 = RING_OFF + BM * RP;
  float mw[2], m2w[2];
#pragma unroll
  for (int a = 0; a < 2; a++) {
    float s = 0.f;
#pragma unroll
    for (int b = 0; b < 10; b++) s += (T[a][b][0] + T[a][b][1]) + (T[a][b][2] + T[a][b][3]);
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    mw[a] = s * (1.f / 160.f);
    float u = 0.f;
#pragma unroll
    for (int b = 0; b < 10; b++)
#pragma unroll
      for (int r = 0; r < 4; r++) { const float d = T[a][b][r] - mw[a]; u += d * d; }
    u += __shfl_xor(u, 16);
    u += __shfl_xor(u, 32);
    m2w[a] = u;
    if (g == 0) *(f2*)(scr + ((a * 16 + rsel) * 4 + wave) * 2) = f2{mw[a], u};
#pragma unroll
    for (int q = 0; q < 5; q++) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = (half_t)T[a][2 * q + (j >> 2)][j & 3];
      *(h8*)(smem + ST0 + (a * 16 + rsel) * RP + (cbase + q * 8) * 2) = o;
    }
  }
  lds_barrier();
#pragma unroll
  for (int a = 0; a < 2; a++) {
    const f2* pr = (const f2*)(scr + (a * 16 + rsel) * 8);
    const f2 s0 = pr[0], s1 = pr[1], s2 = pr[2], s3 = pr[3];
    const float mean = 0.25f * ((s0[0] + s1[0]) + (s2[0] + s3[0]));
    const float d0 = s0[0] - mean, d1 = s1[0] - mean, d2 = s2[0] - mean, d3 = s3[0] - mean;
    const float m2 = ((s0[1] + s1[1]) + (s2[1] + s3[1])) + 160.f * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
    const float rs = 1.f / (sqrtf(m2 * (1.f / CW)) + p.eps);
#pragma unroll
    for (int q = 0; q < 5; q++) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = (half_t)((T[a][2 * q + (j >> 2)][j & 3] - mean) * rs);
      *(h8*)(smem + ST1 + (a * 16 + rsel) * RP + (cbase + q * 8) * 2) = o;
    }
  }
  auto flush_rows = [&](int base, half_t* dst, int ld) {  // 2560 16-B chunks, 80 per row: a wave instruction stores 1 KiB of one or two rows
#pragma unroll
    for (int i = 0; i < 10; i++) {
      const int t = tid + 256 * i, row = t / 80, c = t - row * 80;
      *(h8*)(dst + (long long)(m0 + row) * ld + c * 8) = *(const h8*)(smem + base + row * RP + c * 16);
    }
  };
  flush_rows(ST0, p.tok, p.ld_tok);
  lds_barrier();
  flush_rows(ST1, p.ln, p.ld_ln);
}

size_t row_ln_stream_bytes() { return STREAM_BYTES; }
bool row_ln_supported(int C, int64_t M) { return C == CW && M > 0 && M % BM == 0 && M * CW * 2 < (int64_t)0x7fffff00; }

int launch_row_ln_pack(tsd_ctx* ctx, const half_t* W, int ldw, half_t* dst) {
  if (!W || !dst || ldw < CW || ldw % 8) TSD_FAIL(TSD_E_ARG, "row_ln pack: bad weight matrix");
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_row_ln_pack, dim3(ceil_div(STREAM_BYTES / 16, 256)), dim3(256), 0, ctx->stream, W, ldw, dst);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_row_ln(tsd_ctx* ctx, const RowLnArgs& a) {
  if (!row_ln_supported(CW, a.M)) TSD_FAIL(TSD_E_SHAPE, "row_ln: M=%lld unsupported (C = 640, M %% 32 == 0)", (long long)a.M);
  if (!a.A || !a.wstream || !a.bias || !a.tok || !a.ln) TSD_FAIL(TSD_E_ARG, "row_ln: NULL operand");
  if (a.lda % 8 || a.ld_tok % 8 || a.ld_ln % 8 || a.lda < CW || a.ld_tok < CW || a.ld_ln < CW || (a.R && (a.ldr % 8 || a.ldr < CW)))
    TSD_FAIL(TSD_E_SHAPE, "row_ln: misaligned / short pitches");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ROWLN, (int)a.M, CW, CW, 1);
  RowLnK k;
  k.A = a.A; k.lda = a.lda; k.wstream = a.wstream; k.bias = a.bias; k.R = a.R; k.ldr = a.ldr;
  k.tok = a.tok; k.ld_tok = a.ld_tok; k.ln = a.ln; k.ld_ln = a.ld_ln; k.eps = a.eps;
  static unsigned long long attr = 0;  // one bit per device
  if (!((attr >> (ctx->device & 63)) & 1)) {
    HIP_TRY(hipFuncSetAttribute((const void*)row_ln_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr |= 1ull << (ctx->device & 63);
  }
  hipLaunchKernelGGL(row_ln_kernel, dim3((unsigned)(a.M / BM)), dim3(256), LDS_BYTES, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// Debug / bench entry: the fused kernel against GEMM + LayerNorm on synthetic device data (M rows of 640).
// mode 0: time the fused kernel; 1: time the two-launch path; 2: run both once and report max |difference| of the two
// outputs pairs in ms[0] (tok) and ms[1] (ln).
extern "C" int tsd_debug_row_ln_bench(tsd_ctx* ctx, int M, int with_residual, int mode, int iters, float* ms) {
  if (!ctx || !ms || iters <= 0 || !row_ln_supported(CW, M)) TSD_FAIL(TSD_E_ARG, "row_ln_bench: bad argument");
  HIP_TRY(hipSetDevice(ctx->device));
  const int64_t n = (int64_t)M * CW;
  TSD_TRY(ctx_reserve_arena(ctx, (size_t)n * 2 * 6 + (size_t)n * 4 + (size_t)CW * CW * 2 + STREAM_BYTES + CW * 4 + (1 << 20)));
  ctx->arena.top = 0;
  half_t* A = arena_alloc<half_t>(ctx, n);
  half_t* R = arena_alloc<half_t>(ctx, n);
  half_t* tok0 = arena_alloc<half_t>(ctx, n);
  half_t* ln0 = arena_alloc<half_t>(ctx, n);
  half_t* tok1 = arena_alloc<half_t>(ctx, n);
  half_t* ln1 = arena_alloc<half_t>(ctx, n);
  float* tmp = arena_alloc<float>(ctx, n);
  half_t* W = arena_alloc<half_t>(ctx, (int64_t)CW * CW);
  half_t* ws = arena_alloc<half_t>(ctx, STREAM_BYTES / 2);
  float* bias = arena_alloc<float>(ctx, CW);
  if (!A || !R || !tok0 || !ln0 || !tok1 || !ln1 || !tmp || !W || !ws || !bias) TSD_FAIL(TSD_E_ALLOC, "row_ln_bench: arena");
  TSD_TRY(launch_fill_uniform(ctx, tmp, n, 1, 21, 2.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)n, A, (int)n, 1));
  TSD_TRY(launch_fill_uniform(ctx, tmp, n, 1, 22, 2.f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, (int)n, R, (int)n, 1));
  TSD_TRY(launch_fill_uniform(ctx, tmp, (int64_t)CW * CW, 1, 23, 0.08f));
  TSD_TRY(launch_f32_to_f16_rows(ctx, tmp, 1, CW * CW, W, CW * CW, 1));
  TSD_TRY(launch_fill_uniform(ctx, bias, CW, 1, 24, 0.2f));
  TSD_TRY(launch_row_ln_pack(ctx, W, CW, ws));
  RowLnArgs ra;
  ra.A = A; ra.lda = CW; ra.wstream = ws; ra.bias = bias; ra.R = with_residual ? R : nullptr; ra.ldr = CW;
  ra.tok = tok0; ra.ld_tok = CW; ra.ln = ln0; ra.ld_ln = CW; ra.M = M; ra.eps = 1e-5f;
  auto two_launch = [&]() -> int {
    GemmArgs g;
    g.A0 = A; g.lda0 = CW; g.Wt = W; g.ldw = CW; g.M = M; g.N = CW; g.K = CW; g.batch = 1;
    g.bias = bias; g.epi = EPI_BIAS_N;
    if (with_residual) { g.R = R; g.ldr = CW; g.epi |= EPI_RESIDUAL; }
    g.C = tok1; g.ldc = CW;
    int r = launch_gemm(ctx, g);
    if (r != TSD_OK) return r;
    return launch_layernorm(ctx, tok1, M, CW, CW, 1e-5f, ln1, CW, nullptr);
  };
  int r = TSD_OK;
  if (mode == 2) {
    r = launch_row_ln(ctx, ra);
    if (r == TSD_OK) r = two_launch();
    if (r != TSD_OK) return r;
    std::vector<half_t> h0((size_t)n), h1((size_t)n);
    for (int which = 0; which < 2; which++) {
      HIP_TRY(hipMemcpyAsync(h0.data(), which ? ln0 : tok0, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipMemcpyAsync(h1.data(), which ? ln1 : tok1, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      float mx = 0.f;
      for (int64_t i = 0; i < n; i++) mx = std::max(mx, fabsf((float)h0[(size_t)i] - (float)h1[(size_t)i]));
      ms[which] = mx;
    }
  } else {
    for (int pass = 0; pass < 2; pass++) {
      if (pass == 1) HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
      for (int i = 0; i < (pass ? iters : 2) && r == TSD_OK; i++) r = mode == 0 ? launch_row_ln(ctx, ra) : two_launch();
    }
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, ctx->ev0, ctx->ev1));
    *ms = t / iters;
  }
  ctx->arena.top = 0;
  return r;
}
