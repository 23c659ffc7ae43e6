// kernels_norm.hip - GroupNorm (+SiLU) and LayerNorm on NHWC fp16 activations (HBM-bound).
//
// Replaces `GroupNorm.forward` helpers/utils.mojo:1845-1885 (+ `sum/mean/std` :1360-1380),
// `SiLU.forward` :1892-1902 when fused, and `LayerNorm` :2052-2061 (build semantics App.A D8).
// Formula kept literal (App.A D12): y = (x - mu) / (sigma + eps) * gamma, population sigma,
// eps added to sigma, scalar gamma, no beta.
//
// GroupNorm is an apply pass whose blocks each finish the statistics in double from per-(sample, slab, group)
// (sum, sumsq) partials and stream the tensor once.  The partials normally come for free from the epilogue of the
// GEMM/conv that produced the tensor (EPI_GNSTATS, one 32-row slab per epilogue pass - kernels_gemm.hip); otherwise
// (channel-concat inputs, very large images) from `k_gn_partial` here: coalesced 16-B NHWC loads, 4 in flight per
// thread, fp32 accumulation, no atomics.  The source may be the channel-concat of two tensors (UNet skip
// connections, diffusion.mojo:253-270) so the concat is never materialised for the normalised branch.
// LayerNorm: one group of LPR lanes per row for the UNet widths (k_layernorm_grp), one wave per row otherwise.
#include <stdlib.h>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int GN_MAX_CPT = 2;  // chunks (of 8 channels) per thread: C <= 4096
constexpr int GN_UNROLL = 4;   // independent 16-B loads in flight per thread

struct GnK {
  const half_t* x0; const half_t* x1;
  int ld0, ld1, C0, C, HW, G, cpg;
  int nslab, slab_pixels;
  float* partial;  // [B][nslab][G][2]  (sum, sumsq) per slab
  // composite statistics (comb > 1): the norm's group g is the sum of `comb` consecutive FINE groups of the producers' partials -
  // fine groups [0, G0) from `partial` ([B][nslab][G0][2], the first concat source), [G0, G0 + G1) from `partial1`
  const float* partial1; int G0, G1, comb;
  float* stats;    // [B][G][2]         (mean, gamma/(sigma+eps))
  float eps, gamma;
  int silu;
  int stats_ready;  // stats[] already holds the finished (mean, scale) pairs (k_gn_finalize ran)
  half_t* y; int ldy;
  int apply_pixels;
  const float* aw; const float* ab; int torch_rstd;  // NormAffine extension (common.h); aw == ab == nullptr && !torch_rstd: reference
};

__device__ __forceinline__ h8 gn_load(const GnK& p, int64_t pixg, int ch) {
  const int c = ch * 8;
  if (c < p.C0) return *(const h8*)(p.x0 + pixg * p.ld0 + c);
  return *(const h8*)(p.x1 + pixg * p.ld1 + (c - p.C0));
}

// thread -> (pixel lane pl, chunk column ch0): a thread always handles the same 8 (or 16) channels and
// strides over pixels, so per-channel sums stay in registers.
struct GnMap { int nch, PL, pl, ch0; bool active; };
__device__ __forceinline__ GnMap gn_map(int C, int tid) {
  GnMap m;
  m.nch = C >> 3;
  if (m.nch <= 256) { m.PL = 256 / m.nch; m.pl = tid / m.nch; m.ch0 = tid - m.pl * m.nch; m.active = tid < m.PL * m.nch; }
  else { m.PL = 1; m.pl = 0; m.ch0 = tid; m.active = true; }
  return m;
}

__global__ __launch_bounds__(256) void k_gn_partial(const GnK p) {
  extern __shared__ __attribute__((aligned(16))) char smem_gn[];
  float* red = (float*)smem_gn;  // [PL][2][C]
  const int tid = threadIdx.x;
  const GnMap m = gn_map(p.C, tid);
  const int b = blockIdx.y, slab = blockIdx.x;
  float a1[GN_MAX_CPT][8], a2[GN_MAX_CPT][8];
#pragma unroll
  for (int q = 0; q < GN_MAX_CPT; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) a1[q][j] = a2[q][j] = 0.f;
  const int p_begin = slab * p.slab_pixels;
  const int p_end = min(p.HW, p_begin + p.slab_pixels);
  if (m.active) {
    for (int pix0 = p_begin + m.pl; pix0 < p_end; pix0 += m.PL * GN_UNROLL) {
#pragma unroll
      for (int q = 0; q < GN_MAX_CPT; q++) {
        const int ch = m.ch0 + q * 256;
        if (ch < m.nch) {
          h8 v[GN_UNROLL];
#pragma unroll
          for (int u = 0; u < GN_UNROLL; u++) {
            const int pix = pix0 + u * m.PL;
            v[u] = pix < p_end ? gn_load(p, (int64_t)b * p.HW + pix, ch) : h8{0, 0, 0, 0, 0, 0, 0, 0};
          }
#pragma unroll
          for (int u = 0; u < GN_UNROLL; u++)
#pragma unroll
            for (int j = 0; j < 8; j++) {
              const float f = (float)v[u][j];
              a1[q][j] += f;
              a2[q][j] += f * f;
            }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < GN_MAX_CPT; q++) {
      const int ch = m.ch0 + q * 256;
      if (ch < m.nch) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          red[(m.pl * 2 + 0) * p.C + ch * 8 + j] = a1[q][j];
          red[(m.pl * 2 + 1) * p.C + ch * 8 + j] = a2[q][j];
        }
      }
    }
  }
  __syncthreads();
  // fixed-order (deterministic) reduction: pixel lanes, then the channels of each group
  for (int g = tid; g < p.G; g += 256) {
    float t1 = 0.f, t2 = 0.f;
    for (int c = g * p.cpg; c < (g + 1) * p.cpg; c++)
      for (int l = 0; l < m.PL; l++) {
        t1 += red[(l * 2 + 0) * p.C + c];
        t2 += red[(l * 2 + 1) * p.C + c];
      }
    float* o = p.partial + (((int64_t)b * p.nslab + slab) * p.G + g) * 2;
    o[0] = t1;
    o[1] = t2;
  }
}

// Finish the statistics of 32 groups [g0, g0+32) of sample b from the slab partials: 8 lanes per group stride over the
// slabs, fixed-order butterfly in double => deterministic.  Writes (mean, gamma/(sigma+eps)) pairs to out[2*g..].
__device__ __forceinline__ void gn_finish_groups(const GnK& p, int b, int g0, int tid, float* out) {
  const int g = g0 + (tid >> 3), sl = tid & 7;
  double t1 = 0.0, t2 = 0.0;
  if (g < p.G && p.comb <= 1 && !p.partial1) {
    // a lane's slab partials all in flight at once (16 per batch = 128 slabs, the 64x64 level): with four per batch the loop
    // exposed one L2 / Infinity-Cache round trip per four slabs - 7-8 us for a kernel that reads 32 KB per sample (round 4).
    // Same values added in the same order: bitwise the old result.
    for (int s0 = sl; s0 < p.nslab; s0 += 128) {
      f2 v[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const int s = s0 + 8 * i;
        v[i] = s < p.nslab ? *(const f2*)(p.partial + (((int64_t)b * p.nslab + s) * p.G + g) * 2) : f2{0.f, 0.f};
      }
#pragma unroll
      for (int i = 0; i < 16; i++) { t1 += (double)v[i][0]; t2 += (double)v[i][1]; }
    }
  } else if (g < p.G)
#pragma unroll 4
    for (int s = sl; s < p.nslab; s += 8) {
      {
        for (int k = 0; k < p.comb; k++) {  // fixed order: deterministic
          const int f = g * p.comb + k;
          const float* o = f < p.G0 ? p.partial + (((int64_t)b * p.nslab + s) * p.G0 + f) * 2
                                    : p.partial1 + (((int64_t)b * p.nslab + s) * p.G1 + (f - p.G0)) * 2;
          t1 += (double)o[0];
          t2 += (double)o[1];
        }
      }
    }
#pragma unroll
  for (int o = 4; o > 0; o >>= 1) {
    t1 += __shfl_xor(t1, o);
    t2 += __shfl_xor(t2, o);
  }
  if (g < p.G && sl == 0) {
    const double n = (double)p.cpg * (double)p.HW;
    const double mu = t1 / n;
    double var = t2 / n - mu * mu;
    if (var < 0.0) var = 0.0;
    out[2 * g] = (float)mu;
    // eps added to sigma (helpers/utils.mojo:1871-1873) ; torch_rstd (extension): eps added to the variance
    out[2 * g + 1] = p.torch_rstd ? (float)((double)p.gamma / sqrt(var + (double)p.eps))
                                  : (float)((double)p.gamma / (sqrt(var) + (double)p.eps));
  }
}

// Producer-emitted partials of a LARGE image (VAE: 512 ... 8192 slabs of 32 rows per sample): 64 blocks per sample each add
// nslab/64 consecutive slabs per group - fixed order, double accumulation - into a 64-"slab" table the apply blocks re-reduce
// themselves.  Replaces the statistics pass over the tensor (k_gn_partial: one more read of up to 537 MB) by a read of a few MB.
__global__ __launch_bounds__(256) void k_gn_prereduce(const float* __restrict__ part, int nslab, int G, float* __restrict__ out) {
  const int b = blockIdx.y, c = blockIdx.x, nchunk = gridDim.x;
  const int s0 = (int)(((int64_t)c * nslab) / nchunk), s1 = (int)(((int64_t)(c + 1) * nslab) / nchunk);
  // thread -> (group g, lane l of LN lanes striding the chunk's slabs); 2 floats per (slab, group)
  const int LN = 256 / G > 0 ? 256 / G : 1;
  __shared__ double red[256][2];
  for (int g0 = 0; g0 < G; g0 += 256) {
    const int g = g0 + (int)threadIdx.x % (G < 256 ? G : 256), l = (int)threadIdx.x / (G < 256 ? G : 256);
    double t1 = 0.0, t2 = 0.0;
    if (g < G && l < LN)
      for (int sidx = s0 + l; sidx < s1; sidx += LN) {
        const float* o = part + (((int64_t)b * nslab + sidx) * G + g) * 2;
        t1 += (double)o[0];
        t2 += (double)o[1];
      }
    red[threadIdx.x][0] = t1; red[threadIdx.x][1] = t2;
    __syncthreads();
    if (l == 0 && g < G) {
      for (int k = 1; k < LN; k++) { t1 += red[threadIdx.x + k * G][0]; t2 += red[threadIdx.x + k * G][1]; }  // fixed order
      float* o = out + (((int64_t)b * nchunk + c) * G + g) * 2;
      o[0] = (float)t1; o[1] = (float)t2;
    }
    __syncthreads();
  }
}

// Separate finalize launch, used when (slabs x groups) is large: then every apply block re-reducing the partials would
// read as much as its payload (128 slabs x 32 groups = 32 KB per 30 KB of pixels at 320x64^2; 10x that for the output
// layer's 320 groups).  One block per (32 groups, sample).
__global__ __launch_bounds__(256) void k_gn_finalize(const GnK p) {
  gn_finish_groups(p, blockIdx.y, blockIdx.x * 32, threadIdx.x, p.stats + (int64_t)blockIdx.y * p.G * 2);
}

template <bool AFFINE>
__global__ __launch_bounds__(256) void k_gn_apply(const GnK p) {
  extern __shared__ __attribute__((aligned(16))) char smem_gn[];
  float* st = (float*)smem_gn;  // [G][2] (mean, gamma/(sigma+eps))
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  if (p.stats_ready) {  // finished by k_gn_finalize
    for (int i = tid; i < 2 * p.G; i += 256) st[i] = p.stats[(int64_t)b * p.G * 2 + i];
  } else {  // few slabs: every block finishes the statistics itself (a few KB of partials from L2), no extra launch
    for (int g0 = 0; g0 < p.G; g0 += 32) gn_finish_groups(p, b, g0, tid, st);
  }
  __syncthreads();
  const GnMap m = gn_map(p.C, tid);
  if (!m.active) return;
  float mu[GN_MAX_CPT][8], ri[GN_MAX_CPT][8];
#pragma unroll
  for (int q = 0; q < GN_MAX_CPT; q++) {
    const int ch = m.ch0 + q * 256;
    // group of each of the 8 channels: ONE integer division per chunk (a 32-bit division is ~40 instructions, and this
    // prologue runs in every block for a payload of only 8 pixels per thread), then a running remainder
    int g = ch < m.nch ? (ch * 8) / p.cpg : 0;
    int rem = ch < m.nch ? ch * 8 - g * p.cpg : 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      mu[q][j] = st[2 * g];
      ri[q][j] = st[2 * g + 1];
      if (++rem == p.cpg && ch < m.nch) { rem = 0; ++g; }
    }
  }
  float sh[AFFINE ? GN_MAX_CPT : 1][8];  // AFFINE: per-channel weight folded into ri, bias kept as a shift
  if (AFFINE) {
#pragma unroll
    for (int q = 0; q < GN_MAX_CPT; q++) {
      const int ch = m.ch0 + q * 256;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int c = ch < m.nch ? ch * 8 + j : 0;
        if (p.aw) ri[q][j] *= p.aw[c];
        sh[q][j] = p.ab ? p.ab[c] : 0.f;
      }
    }
  }
  const int p_begin = blockIdx.x * p.apply_pixels;
  const int p_end = min(p.HW, p_begin + p.apply_pixels);
  for (int pix0 = p_begin + m.pl; pix0 < p_end; pix0 += m.PL * GN_UNROLL) {
#pragma unroll
    for (int q = 0; q < GN_MAX_CPT; q++) {
      const int ch = m.ch0 + q * 256;
      if (ch < m.nch) {
        h8 v[GN_UNROLL];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; u++) {
          const int pix = pix0 + u * m.PL;
          if (pix < p_end) v[u] = gn_load(p, (int64_t)b * p.HW + pix, ch);
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; u++) {
          const int pix = pix0 + u * m.PL;
          if (pix < p_end) {
            h8 o;
#pragma unroll
            for (int j = 0; j < 8; j++) {
              float f = ((float)v[u][j] - mu[q][j]) * ri[q][j];
              if (AFFINE) f += sh[q][j];
              if (p.silu) f = f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f));  // x*sigmoid(x): v_exp + v_rcp, no IEEE division
              o[j] = (half_t)f;
            }
            *(h8*)(p.y + ((int64_t)b * p.HW + pix) * p.ldy + ch * 8) = o;
          }
        }
      }
    }
  }
}

int launch_groupnorm(tsd_ctx* ctx, const NormSrc& src, int B, int HW, int C, int groups, float eps, float gamma,
                     int silu, half_t* y, int ldy, const float* pre_part, int pre_nslab, const NormAffine* aff,
                     const GnComposite* comp) {
  if (C % 8 || C % groups || C > 256 * 8 * GN_MAX_CPT)
    TSD_FAIL(TSD_E_SHAPE, "groupnorm: C=%d groups=%d unsupported", C, groups);
  const int C0 = src.x1 ? src.C0 : C;
  if (C0 % 8 || src.ld0 % 8 || (src.x1 && src.ld1 % 8) || ldy % 8) TSD_FAIL(TSD_E_SHAPE, "groupnorm: pitches must be multiples of 8");
  GnK k;
  k.x0 = src.x0; k.x1 = src.x1; k.ld0 = src.ld0; k.ld1 = src.ld1; k.C0 = C0; k.C = C; k.HW = HW; k.G = groups;
  k.cpg = C / groups;
  const int nch = C / 8, PL = nch <= 256 ? 256 / nch : 1;
  // stats pass: 2*GN_UNROLL pixels per thread, at most 512 slabs per sample ; apply pass: GN_UNROLL pixels per thread
  k.slab_pixels = std::max(2 * GN_UNROLL * PL, ceil_div(HW, 64));  // <= 64 slabs: every apply block re-reduces them
  k.nslab = ceil_div(HW, k.slab_pixels);
  k.apply_pixels = (ctx->opt.gn_apply_mult > 0 ? ctx->opt.gn_apply_mult : 2) * GN_UNROLL * PL;
  // statistics already emitted by the producer's epilogue (EPI_GNSTATS, same [B][nslab][G][2] layout): no partial pass
  // composite: the statistics are sums of the producers' finer-grained partials (two concat sources, or one source emitted for a
  // finer grouping) - no statistics pass over the concatenated tensor
  const bool composite = comp && comp->part0 && comp->comb >= 1 && comp->nslab > 0 && comp->nslab <= 256 &&
                         (comp->G0 + (comp->part1 ? comp->G1 : 0)) == groups * comp->comb;
  if (composite) { pre_part = comp->part0; pre_nslab = comp->nslab; }
  const bool have_stats = pre_part != nullptr && pre_nslab > 0;
  const bool prereduce = !composite && have_stats && pre_nslab > 256 && groups <= 256;  // large images (VAE): 64 chunks per sample first
  if (have_stats && !prereduce) { k.partial = const_cast<float*>(pre_part); k.nslab = pre_nslab; }
  else if (prereduce) { k.partial = arena_alloc<float>(ctx, (int64_t)B * 64 * groups * 2); k.nslab = 64; }
  else k.partial = arena_alloc<float>(ctx, (int64_t)B * k.nslab * groups * 2);
  k.stats = arena_alloc<float>(ctx, (int64_t)B * groups * 2);
  if (!k.partial || !k.stats) TSD_FAIL(TSD_E_ALLOC, "groupnorm: workspace exhausted");
  k.eps = eps; k.gamma = gamma; k.silu = silu; k.y = y; k.ldy = ldy;
  k.partial1 = nullptr; k.G0 = groups; k.G1 = 0; k.comb = 1;
  if (composite) { k.partial1 = comp->part1; k.G0 = comp->G0; k.G1 = comp->part1 ? comp->G1 : 0; k.comb = comp->G0 == groups && !comp->part1 ? 1 : comp->comb; }
  k.aw = aff ? aff->w : nullptr; k.ab = aff ? aff->b : nullptr; k.torch_rstd = aff ? aff->torch_rstd : 0;
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_GROUPNORM, B * HW, C, 0, 1);
  k.stats_ready = (int64_t)k.nslab * groups >= ctx->opt.gn_finalize_min ? 1 : 0;
  prof.kernels = 1 + ((!have_stats || prereduce) ? 1 : 0) + (k.stats_ready ? 1 : 0);
  if (!have_stats) {
    hipLaunchKernelGGL(k_gn_partial, dim3(k.nslab, B), dim3(256), (size_t)PL * 2 * C * sizeof(float), ctx->stream, k);
    HIP_TRY(hipGetLastError());
  } else if (prereduce) {
    hipLaunchKernelGGL(k_gn_prereduce, dim3(64, B), dim3(256), 0, ctx->stream, pre_part, pre_nslab, groups, k.partial);
    HIP_TRY(hipGetLastError());
  }
  if (k.stats_ready) {
    hipLaunchKernelGGL(k_gn_finalize, dim3(ceil_div(groups, 32), B), dim3(256), 0, ctx->stream, k);
    HIP_TRY(hipGetLastError());
  }
  if (k.aw || k.ab)
    hipLaunchKernelGGL(k_gn_apply<true>, dim3(ceil_div(HW, k.apply_pixels), B), dim3(256), (size_t)2 * groups * sizeof(float),
                       ctx->stream, k);
  else
    hipLaunchKernelGGL(k_gn_apply<false>, dim3(ceil_div(HW, k.apply_pixels), B), dim3(256), (size_t)2 * groups * sizeof(float),
                       ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

int launch_gn_finalize(tsd_ctx* ctx, const float* partial, int nslab, int B, int HW, int C, int groups, float eps, float gamma,
                       float* stats) {
  if (!partial || nslab <= 0 || !stats || C % groups) TSD_FAIL(TSD_E_ARG, "gn_finalize: bad argument");
  if (!ctx->launch()) return TSD_OK;
  GnK k = {};
  k.C = C; k.HW = HW; k.G = groups; k.cpg = C / groups; k.nslab = nslab;
  k.partial = const_cast<float*>(partial); k.stats = stats; k.eps = eps; k.gamma = gamma; k.torch_rstd = 0;
  ProfScope prof(ctx, KC_GROUPNORM, B * groups, nslab, 0, 1);
  hipLaunchKernelGGL(k_gn_finalize, dim3(ceil_div(groups, 32), B), dim3(256), 0, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// statistics only (no apply pass): partial sums over pixel slabs, then the finalize above -> stats [B][groups][2]
int launch_gn_stats(tsd_ctx* ctx, const half_t* x, int ld, int B, int HW, int C, int groups, float eps, float gamma, float* stats) {
  if (C % 8 || C % groups || C > 256 * 8 * GN_MAX_CPT || ld % 8) TSD_FAIL(TSD_E_SHAPE, "gn_stats: C=%d groups=%d unsupported", C, groups);
  GnK k = {};
  k.x0 = x; k.ld0 = ld; k.C0 = C; k.C = C; k.HW = HW; k.G = groups; k.cpg = C / groups;
  const int nch = C / 8, PL = nch <= 256 ? 256 / nch : 1;
  k.slab_pixels = std::max(2 * GN_UNROLL * PL, ceil_div(HW, 64));
  k.nslab = ceil_div(HW, k.slab_pixels);
  k.partial = arena_alloc<float>(ctx, (int64_t)B * k.nslab * groups * 2);
  if (!k.partial) TSD_FAIL(TSD_E_ALLOC, "gn_stats: workspace exhausted");
  k.stats = stats; k.eps = eps; k.gamma = gamma;
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_GROUPNORM, B * HW, C, 0, 1);
  prof.kernels = 2;
  hipLaunchKernelGGL(k_gn_partial, dim3(k.nslab, B), dim3(256), (size_t)PL * 2 * C * sizeof(float), ctx->stream, k);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_gn_finalize, dim3(ceil_div(groups, 32), B), dim3(256), 0, ctx->stream, k);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- LayerNorm over the last dim of (rows, C): one wave per row, data held in registers ----
constexpr int LN_MAX_CH = 4;  // C <= 2048
struct LnAff { const float* w; const float* b; int torch_rstd; };
template <bool AFFINE>
__global__ __launch_bounds__(256) void k_layernorm(const half_t* __restrict__ x, int64_t rows, int C, int ldx, float eps,
                                                   half_t* __restrict__ y, int ldy, LnAff aff) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = C >> 3;
  const half_t* xr = x + row * ldx;
  h8 v[LN_MAX_CH];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAX_CH; q++) {
    const int ch = lane + q * 64;
    if (ch < nch) {
      v[q] = *(const h8*)(xr + ch * 8);
#pragma unroll
      for (int j = 0; j < 8; j++) s += (float)v[q][j];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAX_CH; q++) {
    const int ch = lane + q * 64;
    if (ch < nch) {
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float d = (float)v[q][j] - mu;
        ss += d * d;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float r = (AFFINE && aff.torch_rstd) ? rsqrtf(ss / (float)C + eps) : 1.f / (sqrtf(ss / (float)C) + eps);
  half_t* yr = y + row * ldy;
#pragma unroll
  for (int q = 0; q < LN_MAX_CH; q++) {
    const int ch = lane + q * 64;
    if (ch < nch) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        float f = ((float)v[q][j] - mu) * r;
        if (AFFINE) f = f * (aff.w ? aff.w[ch * 8 + j] : 1.f) + (aff.b ? aff.b[ch * 8 + j] : 0.f);
        o[j] = (half_t)f;
      }
      *(h8*)(yr + ch * 8) = o;
    }
  }
}

// UNet token widths (C = 40 * LPR, LPR a power of two <= 32): LPR lanes share a row, 5 chunks of 8 channels each, so a
// wave normalises 64 / LPR rows at once with every lane busy and five independent 16-B loads in flight per lane
// (the one-wave-per-row kernel above leaves 24 of 64 lanes idle at C = 320 and moves 640 B per wave).
template <int LPR, bool AFFINE>
__global__ __launch_bounds__(256) void k_layernorm_grp(const half_t* __restrict__ x, int64_t rows, int ldx, float eps,
                                                       half_t* __restrict__ y, int ldy, LnAff aff) {
  constexpr int RPW = 64 / LPR, C = 40 * LPR;
  const int lane = threadIdx.x & 63, sub = lane % LPR;
  const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool ok = row < rows;
  const half_t* xr = x + (ok ? row : rows - 1) * ldx;
  h8 v[5];
#pragma unroll
  for (int q = 0; q < 5; q++) v[q] = *(const h8*)(xr + (sub + q * LPR) * 8);
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 5; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) s += (float)v[q][j];
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < 5; q++)
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const float d = (float)v[q][j] - mu;
      ss += d * d;
    }
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  const float r = (AFFINE && aff.torch_rstd) ? rsqrtf(ss / (float)C + eps) : 1.f / (sqrtf(ss / (float)C) + eps);
  if (!ok) return;
  half_t* yr = y + row * ldy;
#pragma unroll
  for (int q = 0; q < 5; q++) {
    h8 o;
    const int c0 = (sub + q * LPR) * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      float f = ((float)v[q][j] - mu) * r;
      if (AFFINE) f = f * (aff.w ? aff.w[c0 + j] : 1.f) + (aff.b ? aff.b[c0 + j] : 0.f);
      o[j] = (half_t)f;
    }
    *(h8*)(yr + c0) = o;
  }
}

template <bool AFFINE>
static void launch_layernorm_t(tsd_ctx* ctx, const half_t* x, int64_t rows, int C, int ldx, float eps, half_t* y, int ldy,
                               LnAff a) {
  if (C == 320 || C == 640 || C == 1280) {
    const int lpr = C / 40, rows_per_block = 4 * (64 / lpr);
    const dim3 grid((unsigned)((rows + rows_per_block - 1) / rows_per_block));
    if (lpr == 8) hipLaunchKernelGGL((k_layernorm_grp<8, AFFINE>), grid, dim3(256), 0, ctx->stream, x, rows, ldx, eps, y, ldy, a);
    else if (lpr == 16) hipLaunchKernelGGL((k_layernorm_grp<16, AFFINE>), grid, dim3(256), 0, ctx->stream, x, rows, ldx, eps, y, ldy, a);
    else hipLaunchKernelGGL((k_layernorm_grp<32, AFFINE>), grid, dim3(256), 0, ctx->stream, x, rows, ldx, eps, y, ldy, a);
  } else {
    hipLaunchKernelGGL(k_layernorm<AFFINE>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, x, rows, C, ldx, eps,
                       y, ldy, a);
  }
}

int launch_layernorm(tsd_ctx* ctx, const half_t* x, int64_t rows, int C, int ldx, float eps, half_t* y, int ldy,
                     const NormAffine* aff) {
  if (C % 8 || C > 64 * 8 * LN_MAX_CH) TSD_FAIL(TSD_E_SHAPE, "layernorm: C=%d unsupported", C);
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_LAYERNORM, (int)rows, C, 0, 1);
  LnAff a{aff ? aff->w : nullptr, aff ? aff->b : nullptr, aff ? aff->torch_rstd : 0};
  if (aff) launch_layernorm_t<true>(ctx, x, rows, C, ldx, eps, y, ldy, a);
  else launch_layernorm_t<false>(ctx, x, rows, C, ldx, eps, y, ldy, a);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
