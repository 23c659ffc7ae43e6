// kernels_elementwise.hip - HBM-bound helpers: boundary layout conversion (reference CHW fp32 <->
// device NHWC fp16), op-level fp32 elementwise ops, row softmax, time embedding, the tiny
// M<=16 linears of the time path, weight packing, counter-RNG init and the DDPM update.
#include <atomic>

#include "common.h"

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define GRID1D(n, bs) dim3((unsigned)std::min<int64_t>(((n) + (bs)-1) / (bs), 1 << 20))
#include <algorithm>

// ---- boundary layout conversion ------------------------------------
// [B][C][H][W] fp32 -> [B][H][W][Cdst] fp16; channels >= Cuse are zero (K padding for the
// implicit GEMM); value * scale (Decoder's x/0.18215, vae.mojo:222, folded here).
__global__ void k_chw_to_nhwc(const float* __restrict__ src, int C, int HW, int Cuse, float scale,
                              half_t* __restrict__ dst, int Cdst, int64_t total_chunks) {
  const int cchunks = Cdst >> 3;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks;
       i += (int64_t)gridDim.x * blockDim.x) {
    // pixel fastest so that the strided fp32 reads of neighbouring threads coalesce
    const int64_t pix_total = total_chunks / cchunks;
    const int cc = (int)(i / pix_total);
    const int64_t pixg = i - (int64_t)cc * pix_total;  // b*HW + pix
    const int64_t b = pixg / HW;
    const int64_t pix = pixg - b * HW;
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int c = cc * 8 + j;
      v[j] = (c < Cuse) ? (half_t)(src[(b * C + c) * HW + pix] * scale) : (half_t)0.f;
    }
    *(h8*)(dst + pixg * Cdst + cc * 8) = v;
  }
}
int launch_chw_f32_to_nhwc_f16(tsd_ctx* ctx, const float* src, int B, int C, int H, int W, int Cuse, float scale,
                               half_t* dst, int Cdst) {
  if (Cdst % 8) TSD_FAIL(TSD_E_SHAPE, "nhwc channel pitch %d not a multiple of 8", Cdst);
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)B * H * W * (Cdst / 8);
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_chw_to_nhwc, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, C, H * W, Cuse, scale, dst,
                     Cdst, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// im2col rows of a 3x3 / stride 1 / pad 1 convolution over a C-channel (C <= 7) fp32 CHW image: one thread per 16-B chunk of a
// 64-wide row; column t*C + c = tap t = (kh, kw), channel c.  The 4-channel latent padded to 64 channels cost the input
// convolution nine K tiles of which 1/16 carried data (diffusion.mojo:236 `Conv2D(4, 320, 3)`).
__global__ void k_chw_to_im2col3x3(const float* __restrict__ src, int C, int H, int W, half_t* __restrict__ dst, int64_t total_chunks, float scale) {
  const int HW = H * W;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
    const int cc = (int)(i & 7);
    const int64_t pixg = i >> 3;
    const int64_t b = pixg / HW;
    const int pix = (int)(pixg - b * HW), y = pix / W, x = pix - y * W;
    h8 v;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int col = cc * 8 + j, t = col / C, c = col - t * C;
      const int iy = y + t / 3 - 1, ix = x + t % 3 - 1;
      const bool ok = t < 9 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
      v[j] = ok ? (half_t)(src[(b * C + c) * HW + iy * W + ix] * scale) : (half_t)0.f;
    }
    *(h8*)(dst + pixg * 64 + cc * 8) = v;
  }
}
int launch_chw_f32_to_im2col3x3_f16(tsd_ctx* ctx, const float* src, int B, int C, int H, int W, half_t* dst, float scale) {
  if (C <= 0 || 9 * C > 64) TSD_FAIL(TSD_E_SHAPE, "im2col: %d channels do not fit a 64-wide row", C);
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)B * H * W * 8;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_chw_to_im2col3x3, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, C, H, W, dst, total, scale);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
// packed conv3x3 weights [O][9][Ipad] fp16 -> [O][64]: column t*C + c = w[o][t][c], zero beyond 9*C
__global__ void k_pack_im2col_w(const half_t* __restrict__ w, int O, int Ipad, int C, half_t* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O * 64) return;
  const int o = i >> 6, col = i & 63, t = col / C, c = col - t * C;
  dst[i] = t < 9 ? w[((int64_t)o * 9 + t) * Ipad + c] : (half_t)0.f;
}
int launch_pack_im2col_w(tsd_ctx* ctx, const half_t* w, int O, int Ipad, int C, half_t* dst) {
  if (C <= 0 || 9 * C > 64 || C > Ipad) TSD_FAIL(TSD_E_SHAPE, "im2col weights: %d channels", C);
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_pack_im2col_w, dim3((O * 64 + 255) / 256), dim3(256), 0, ctx->stream, w, O, Ipad, C, dst);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// [N][K] fp16 -> K-tile-major [K/64][N][64] (16-B chunks)
__global__ void k_pack_tile_major(const half_t* __restrict__ w, int N, int K, half_t* __restrict__ dst, int64_t total_chunks) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total_chunks; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i & 7);
    const int64_t r = i >> 3;       // kt * N + n
    const int64_t kt = r / N, n = r - kt * N;
    *(h8*)(dst + i * 8) = *(const h8*)(w + n * K + kt * 64 + c * 8);
  }
}
int launch_pack_tile_major(tsd_ctx* ctx, const half_t* w, int N, int K, half_t* dst) {
  if (K % 64) TSD_FAIL(TSD_E_SHAPE, "tile-major pack: K=%d", K);
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)N * K / 8;
  hipLaunchKernelGGL(k_pack_tile_major, GRID1D(total, 256), dim3(256), 0, ctx->stream, w, N, K, dst, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// Linear layer folded into the 1x1 convolution behind it (model_check_ready; the attention blocks that run op by op): one thread per folded
// weight, fp32 sum over the C inner channels in index order (deterministic), one rounding to fp16.  8.4 G multiply-adds at C = 1280: a few ms, once.
__global__ __launch_bounds__(256) void k_fold_linear_conv1x1(const half_t* __restrict__ wo, int ldo, const float* __restrict__ bo, const half_t* __restrict__ w2, int ld2,
                                                             const float* __restrict__ b2, int C, int K2, half_t* __restrict__ wf, int ldf, float* __restrict__ bf) {
  const int n = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
  const half_t* won = wo + (int64_t)n * ldo;
  if (k < K2) {
    float acc = 0.f;
    for (int j = 0; j < C; j++) acc += (float)won[j] * (float)w2[(int64_t)j * ld2 + k];
    wf[(int64_t)n * ldf + k] = (half_t)acc;
  } else if (k < K2 + C) {
    wf[(int64_t)n * ldf + k] = won[k - K2];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    float acc = 0.f;
    if (b2) for (int j = 0; j < C; j++) acc += (float)won[j] * b2[j];
    bf[n] = acc + (bo ? bo[n] : 0.f);
  }
}
int launch_fold_linear_conv1x1(tsd_ctx* ctx, const half_t* wo, int ldo, const float* bo, const half_t* w2, int ld2, const float* b2, int C, int K2,
                               half_t* wf, int ldf, float* bf) {
  if (C <= 0 || K2 <= 0 || ldo < C || ld2 < K2 || ldf < K2 + C) TSD_FAIL(TSD_E_SHAPE, "fold: C=%d K2=%d pitches %d / %d / %d", C, K2, ldo, ld2, ldf);
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_fold_linear_conv1x1, dim3((K2 + C + 255) / 256, C), dim3(256), 0, ctx->stream, wo, ldo, bo, w2, ld2, b2, C, K2, wf, ldf, bf);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- non-finite accounting at the device -> caller exits ------------------------------------------------------------------
// The reference computes in fp32 (helpers/utils.mojo:12-15) and cannot overflow at the path's magnitudes; this path stores
// activations as fp16 (|x| <= 65504).  An overflow turns into inf / NaN that the following norms and GEMMs spread, so it reaches
// the tensors that LEAVE the device: every kernel that produces a caller-visible tensor counts the non-finite values it writes in
// the context's status word, and the synchronisation points of the ABI turn a non-zero count into TSD_E_NONFINITE (runtime.cpp).
__device__ __forceinline__ bool nonfinite_f(float v) { return !(fabsf(v) <= 3.4028234e38f); }  // inf or NaN
__device__ __forceinline__ void nonfinite_report(int* counter, int nbad) {
  if (nbad) atomicAdd(counter, nbad);  // rare path
}

template <class T>
__global__ void k_nhwc_to_chw(const T* __restrict__ src, int C, int HW, int ld, float* __restrict__ dst,
                              int64_t total, int* __restrict__ nonfinite) {
  int nbad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % HW;
    const int64_t bc = i / HW;
    const int c = (int)(bc % C);
    const int64_t b = bc / C;
    const float v = (float)src[(b * HW + pix) * ld + c];
    nbad += nonfinite_f(v);
    dst[i] = v;
  }
  nonfinite_report(nonfinite, nbad);
}
int launch_nhwc_f16_to_chw_f32(tsd_ctx* ctx, const half_t* src, int B, int C, int H, int W, int ld, float* dst) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)B * C * H * W;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_nhwc_to_chw<half_t>, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, C, H * W, ld, dst,
                     total, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
int launch_nhwc_f32_to_chw_f32(tsd_ctx* ctx, const float* src, int B, int C, int H, int W, int ld, float* dst) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)B * C * H * W;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_nhwc_to_chw<float>, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, C, H * W, ld, dst,
                     total, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// rows x cols fp32 -> fp16 [rows_dst][ld_dst], zero padded
__global__ void k_f32_to_f16_rows(const float* __restrict__ src, int64_t rows, int cols, half_t* __restrict__ dst,
                                  int ld, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ld;
    const int c = (int)(i - r * ld);
    dst[i] = (r < rows && c < cols) ? (half_t)src[r * cols + c] : (half_t)0.f;
  }
}
int launch_f32_to_f16_rows(tsd_ctx* ctx, const float* src, int64_t rows, int cols, half_t* dst, int ld_dst,
                           int64_t rows_dst) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = rows_dst * ld_dst;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_f32_to_f16_rows, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, rows, cols, dst, ld_dst,
                     total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
__global__ void k_f16_to_f32_rows(const half_t* __restrict__ src, int cols, int ld, float* __restrict__ dst,
                                  int64_t total, int* __restrict__ nonfinite) {
  int nbad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float v = (float)src[r * ld + c];
    nbad += nonfinite_f(v);
    dst[i] = v;
  }
  nonfinite_report(nonfinite, nbad);
}
int launch_f16_to_f32_rows(tsd_ctx* ctx, const half_t* src, int64_t rows, int cols, int ld_src, float* dst) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = rows * cols;
  hipLaunchKernelGGL(k_f16_to_f32_rows, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, cols, ld_src, dst, total, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- op-level fp32 elementwise ------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }  // helpers/utils.mojo:1898
__device__ __forceinline__ float gelu_f(float x) {                                   // helpers/utils.mojo:1914
  return 0.5f * x * (1.f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__global__ void k_unary(int op, const float* __restrict__ x, int64_t n, float* __restrict__ y, int* __restrict__ nonfinite) {
  int nbad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float v = x[i];
    float r;
    if (op == 0) r = v / (1.f + expf(-v));
    else if (op == 1) r = gelu_f(v);
    else { r = fminf(fmaxf((v + 1.f) * 127.5f, 0.f), 255.f); nbad += nonfinite_f(v); }  // pipeline.mojo:127 (the clamp would swallow a NaN)
    y[i] = r;
  }
  nonfinite_report(nonfinite, nbad);
}
int launch_unary_f32(tsd_ctx* ctx, int op, const float* x, int64_t n, float* y) {
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_unary, GRID1D(n, 256), dim3(256), 0, ctx->stream, op, x, n, y, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

__global__ void k_pad(const float* __restrict__ x, int C, int H, int W, int t, int l, int Ho, int Wo,
                      float* __restrict__ y, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % Wo);
    const int oy = (int)((i / Wo) % Ho);
    const int c = (int)(i / ((int64_t)Wo * Ho));
    const int iy = oy - t, ix = ox - l;
    y[i] = ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? x[((int64_t)c * H + iy) * W + ix] : 0.f;
  }
}
int launch_pad_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int t, int b, int l, int r, float* y) {
  if (!ctx->launch()) return TSD_OK;
  const int Ho = H + t + b, Wo = W + l + r;
  const int64_t total = (int64_t)C * Ho * Wo;
  hipLaunchKernelGGL(k_pad, GRID1D(total, 256), dim3(256), 0, ctx->stream, x, C, H, W, t, l, Ho, Wo, y, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
__global__ void k_upsample(const float* __restrict__ x, int H, int W, float* __restrict__ y, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % (2 * W));
    const int oy = (int)((i / (2 * W)) % (2 * H));
    const int64_t c = i / ((int64_t)4 * W * H);
    y[i] = x[(c * H + (oy >> 1)) * W + (ox >> 1)];
  }
}
int launch_upsample_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, float* y) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)C * H * W * 4;
  hipLaunchKernelGGL(k_upsample, GRID1D(total, 256), dim3(256), 0, ctx->stream, x, H, W, y, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- row softmax (helpers/utils.mojo:411-448 over the key axis, with max subtraction) ------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// causal_period > 0: row r keeps columns <= r % causal_period (upper-triangle mask of helpers/attention.mojo:48-55,
// intended form App.A D7); columns [kept, zero_to) are written as exact zeros (masked scores and K-padding of the
// following P.V GEMM).
template <class T>
__global__ __launch_bounds__(256) void k_softmax_rows(const T* __restrict__ x, int cols, int ldx, T* __restrict__ y,
                                                      int ldy, int causal_period, int zero_to) {
  __shared__ float red[8];
  const int64_t row = blockIdx.x;
  const T* xr = x + row * ldx;
  T* yr = y + row * ldy;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (causal_period > 0) cols = min(cols, (int)(row % causal_period) + 1);
  for (int c = cols + tid; c < zero_to; c += 256) yr[c] = (T)0.f;
  float m = -3.0e38f;
  for (int c = tid; c < cols; c += 256) m = fmaxf(m, (float)xr[c]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
  for (int c = tid; c < cols; c += 256) s += __expf((float)xr[c] - m);
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  s = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.f / s;
  for (int c = tid; c < cols; c += 256) yr[c] = (T)(__expf((float)xr[c] - m) * inv);
}
// fp16 rows of up to 256 * 8 * NV columns (cols % 8 == 0), in place: the row is read ONCE into registers (16-B loads), reduced,
// and written once.  The VAE attention's 4096 x 4096 score matrices (vae.mojo:17-27 via helpers/attention.mojo:46-58) are 268 MB
// per batch of 8: the generic kernel above walks each row three times with 2-byte accesses (0.31 ms of the decode).
template <int NV>
__global__ __launch_bounds__(256) void k_softmax_rows_h8(half_t* __restrict__ x, int cols, int ld) {
  __shared__ float red[8];
  half_t* xr = x + (int64_t)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  h8 v[NV];
  float m = -3.0e38f;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    const int c = (tid + q * 256) * 8;
    if (c < cols) {
      v[q] = *(const h8*)(xr + c);
#pragma unroll
      for (int j = 0; j < 8; j++) m = fmaxf(m, (float)v[q][j]);
    }
  }
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float e[NV][8];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    const int c = (tid + q * 256) * 8;
    if (c < cols) {
#pragma unroll
      for (int j = 0; j < 8; j++) { e[q][j] = __expf((float)v[q][j] - m); s += e[q][j]; }
    }
  }
  s = wave_sum(s);
  if (lane == 0) red[4 + wave] = s;
  __syncthreads();
  s = red[4] + red[5] + red[6] + red[7];
  const float inv = 1.f / s;
#pragma unroll
  for (int q = 0; q < NV; q++) {
    const int c = (tid + q * 256) * 8;
    if (c < cols) {
      h8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = (half_t)(e[q][j] * inv);
      *(h8*)(xr + c) = o;
    }
  }
}
int launch_softmax_rows_f32(tsd_ctx* ctx, const float* x, int64_t rows, int cols, float* y) {
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_softmax_rows<float>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, cols, cols, y, cols, 0, 0);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
int launch_softmax_rows_f16(tsd_ctx* ctx, half_t* x, int64_t rows, int cols, int ld) {
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_SOFTMAX);
  if (cols % 8 == 0 && ld % 8 == 0 && cols <= 256 * 8 * 2) {  // register-resident rows
    if (cols <= 256 * 8) hipLaunchKernelGGL(k_softmax_rows_h8<1>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, cols, ld);
    else hipLaunchKernelGGL(k_softmax_rows_h8<2>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, x, cols, ld);
    HIP_TRY(hipGetLastError());
    return TSD_OK;
  }
  hipLaunchKernelGGL(k_softmax_rows<half_t>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const half_t*)x, cols,
                     ld, x, ld, 0, 0);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
int launch_softmax_rows_f16_causal(tsd_ctx* ctx, half_t* x, int64_t rows, int cols, int ld, int period, int zero_to) {
  if (period <= 0 || zero_to > ld) TSD_FAIL(TSD_E_ARG, "causal softmax: bad period / padding");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_SOFTMAX);
  hipLaunchKernelGGL(k_softmax_rows<half_t>, dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const half_t*)x, cols,
                     ld, x, ld, period, zero_to);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- CLIP: token + position embedding (clip.mojo:17-20, helpers/utils.mojo:2032-2046 intended form App.A D3) ----
__global__ __launch_bounds__(128) void k_clip_embed(const int* __restrict__ tokens, const half_t* __restrict__ table,
                                                    int n_vocab, int D, const float* __restrict__ pos, int T,
                                                    half_t* __restrict__ y) {
  const int64_t row = blockIdx.x;  // b * T + t
  int tok = tokens[row];
  tok = tok < 0 ? 0 : (tok >= n_vocab ? n_vocab - 1 : tok);  // index clamp like Matrix.__getitem__ (helpers/utils.mojo:770-777)
  const half_t* e = table + (int64_t)tok * D;
  const float* pr = pos + (int64_t)(row % T) * D;
  for (int c = threadIdx.x * 8; c < D; c += 128 * 8) {
    const h8 v = *(const h8*)(e + c);
    h8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = (half_t)((float)v[j] + pr[c + j]);
    *(h8*)(y + row * D + c) = o;
  }
}
int launch_clip_embed(tsd_ctx* ctx, const int* tokens, const half_t* table, int n_vocab, int D, const float* pos, int B,
                      int T, half_t* y) {
  if (D % 8) TSD_FAIL(TSD_E_SHAPE, "clip embedding: D=%d", D);
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_clip_embed, dim3(B * T), dim3(128), 0, ctx->stream, tokens, table, n_vocab, D, pos, T, y);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- CLIP: quick-GELU x * sigmoid(1.702 x) in place on fp16 (clip.mojo:49-50, intended form App.A D15) ----
__global__ __launch_bounds__(256) void k_quick_gelu_f16(half_t* __restrict__ x, int64_t n8) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  h8 v = *(const h8*)(x + i * 8);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const float f = (float)v[j];
    v[j] = (half_t)(f / (1.f + __expf(-1.702f * f)));
  }
  *(h8*)(x + i * 8) = v;
}
int launch_quick_gelu_f16(tsd_ctx* ctx, half_t* x, int64_t n) {
  if (n % 8) TSD_FAIL(TSD_E_SHAPE, "quick_gelu: n must be a multiple of 8");
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_quick_gelu_f16, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, ctx->stream, x, n / 8);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- time embedding (helpers/utils.mojo:353-370, build semantics App.A D9) -----------------------
__global__ void k_time_embedding(const float* __restrict__ t, float t_scalar, float* __restrict__ out) {
  const int b = blockIdx.x, i = threadIdx.x;  // 160 threads
  if (i >= 160) return;
  const float f = (float)pow(10000.0, -(double)i / 160.0);
  const float x = f * (t ? t[b] : t_scalar);
  out[b * 320 + i] = (float)cos((double)x);
  out[b * 320 + 160 + i] = (float)sin((double)x);
}
int launch_time_embedding(tsd_ctx* ctx, const float* t, float t_scalar, int B, float* out) {
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_time_embedding, dim3(B), dim3(192), 0, ctx->stream, t, t_scalar, out);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- tiny-M linear (time MLP and the nine 1280->C time projections; SURVEY.md App.D K8) ----------
// y[b][n] = sum_k act(x[b][k]) * W[n][k] + bias[n], B <= 16.  Weights are streamed once (HBM-bound GEMV); act(x) is staged
// in LDS per block.  A block owns 16 output columns, a wave 4 of them with 16 lanes per column: each lane covers K/16 of
// the row with all its 16-B weight loads issued up front, and a column is finished by a 4-step exchange inside its 16
// lanes (the earlier one-column-per-wave layout needed 6 LDS-crossbar steps for every (sample, column) pair: 22 us
// floor per launch whatever the size).
constexpr int SL_MAXB = 16;
constexpr int SL_KIT = 16;  // 16-B weight chunks per lane held in registers: K <= 16 * 128 = 2048 per pass
template <int NB>
__global__ __launch_bounds__(256) void k_small_linear(const float* __restrict__ x, int B, int K, int ldx,
                                                      const half_t* __restrict__ w, int ldw,
                                                      const float* __restrict__ bias, int N, int silu_in,
                                                      float* __restrict__ y, int ldy) {
  extern __shared__ __attribute__((aligned(16))) char smem_sl[];
  float* xs = (float*)smem_sl;  // [B][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the lane's weight chunks of the first (for K <= 2048: the only) pass are requested BEFORE the activations are staged: the
  // two round trips (weights cold from HBM, activations from the previous launch) overlap instead of following each other
  const int col = lane >> 4, sub = lane & 15;  // column within the wave's 4, lane within the column's 16
  const int n = blockIdx.x * 16 + wave * 4 + col;
  const half_t* wr = w + (int64_t)min(n, N - 1) * ldw;
  h8 wv0[SL_KIT];
#pragma unroll
  for (int it = 0; it < SL_KIT; it++) {
    const int k0 = it * 128 + sub * 8;
    wv0[it] = k0 < K ? *(const h8*)(wr + k0) : h8{0, 0, 0, 0, 0, 0, 0, 0};
  }
  // activations -> LDS (with the fused SiLU): batches of 8 independent 16-B loads per thread (K % 4 == 0)
  const int nvec = (B * K) >> 2;
  for (int i0 = tid; i0 < nvec; i0 += 256 * 8) {
    f4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u * 256;
      if (i < nvec) {
        const int e = i * 4, b = e / K, k = e - b * K;
        v[u] = *(const f4*)(x + (int64_t)b * ldx + k);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int i = i0 + u * 256;
      if (i < nvec) {
        if (silu_in) {
#pragma unroll
          for (int j = 0; j < 4; j++) v[u][j] = v[u][j] * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[u][j]));
        }
        *(f4*)(xs + i * 4) = v[u];
      }
    }
  }
  __syncthreads();
  float acc[NB];
#pragma unroll
  for (int b = 0; b < NB; b++) acc[b] = 0.f;
  for (int kbase = 0; kbase < K; kbase += 128 * SL_KIT) {
    h8 wv[SL_KIT];
#pragma unroll
    for (int it = 0; it < SL_KIT; it++) {
      const int k0 = kbase + it * 128 + sub * 8;
      if (kbase == 0) wv[it] = wv0[it];
      else wv[it] = k0 < K ? *(const h8*)(wr + k0) : h8{0, 0, 0, 0, 0, 0, 0, 0};
    }
#pragma unroll
    for (int it = 0; it < SL_KIT; it++) {
      const int k0 = kbase + it * 128 + sub * 8;
      if (k0 < K) {
        const h8 q = wv[it];
#pragma unroll
        for (int b = 0; b < NB; b++)
          if (b < B) {
            const f4 x0 = *(const f4*)(xs + b * K + k0), x1 = *(const f4*)(xs + b * K + k0 + 4);
            acc[b] += (float)q[0] * x0[0] + (float)q[1] * x0[1] + (float)q[2] * x0[2] + (float)q[3] * x0[3] +
                      (float)q[4] * x1[0] + (float)q[5] * x1[1] + (float)q[6] * x1[2] + (float)q[7] * x1[3];
          }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < NB; b++) {
    float s = acc[b];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);  // within the column's 16 lanes
    if (sub == 0 && b < B && n < N) y[(int64_t)b * ldy + n] = s + (bias ? bias[n] : 0.f);
  }
}

int launch_small_linear(tsd_ctx* ctx, const float* x, int B, int K, int ldx, const half_t* w, int ldw, const float* bias,
                        int N, int silu_in, float* y, int ldy) {
  if (B > SL_MAXB) TSD_FAIL(TSD_E_SHAPE, "small_linear: B=%d > %d", B, SL_MAXB);
  if (K % 8 || ldx % 4) TSD_FAIL(TSD_E_SHAPE, "small_linear: K=%d must be a multiple of 8 (ldx=%d of 4)", K, ldx);
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_SMALL_LINEAR, B, N, K, 1);
  const size_t lds = (size_t)B * K * sizeof(float);
  if (lds > 160 * 1024) TSD_FAIL(TSD_E_SHAPE, "small_linear: B*K too large for LDS");
  static std::atomic<unsigned long long> attr_set[5];  // per DEVICE: the attribute is stored per device (one bit each); zero-initialised
  const int slot = B <= 1 ? 0 : B <= 2 ? 1 : B <= 4 ? 2 : B <= 8 ? 3 : 4;
  auto launch = [&](auto fn) -> int {
    if (!((attr_set[slot].load(std::memory_order_relaxed) >> (ctx->device & 63)) & 1)) {
      HIP_TRY(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      attr_set[slot].fetch_or(1ull << (ctx->device & 63), std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(fn, dim3(ceil_div(N, 16)), dim3(256), lds, ctx->stream, x, B, K, ldx, w, ldw, bias, N, silu_in, y, ldy);
    return TSD_OK;
  };
  if (B <= 1) TSD_TRY(launch(k_small_linear<1>));
  else if (B <= 2) TSD_TRY(launch(k_small_linear<2>));
  else if (B <= 4) TSD_TRY(launch(k_small_linear<4>));
  else if (B <= 8) TSD_TRY(launch(k_small_linear<8>));
  else TSD_TRY(launch(k_small_linear<16>));
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- counter RNG (bit-identical to tsd/rng.py and oracle/rng.py) ---------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27; z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return z;
}
__global__ void k_fill_uniform(float* __restrict__ dst, int64_t n, uint64_t base, float bound) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = mix64((uint64_t)i + base);
    const float u = (float)(uint32_t)(h >> 40);                   // 24 bits, exact
    const float v = __fsub_rn(__fmul_rn(u, 1.1920928955078125e-07f), 1.0f);  // u*2^-23 - 1, exact
    dst[i] = __fmul_rn(v, bound);
  }
}
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7)
__device__ __forceinline__ float gelu_erf_f(float x) {
  const float z = fabsf(x) * 0.7071067811865476f;
  const float t = __builtin_amdgcn_rcpf(1.f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float erf_abs = 1.f - poly * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
  return 0.5f * x * (1.f + copysignf(erf_abs, x));
}
__global__ void k_geglu_erf(const half_t* __restrict__ x, int64_t n_pairs8, half_t* __restrict__ out) {
  typedef _Float16 h8v __attribute__((ext_vector_type(8)));
  typedef _Float16 h4v __attribute__((ext_vector_type(4)));
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pairs8; i += (int64_t)gridDim.x * blockDim.x) {
    const h8v v = *(const h8v*)(x + i * 8);  // 4 (a, gate) pairs
    h4v o;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = (half_t)((float)v[2 * j] * gelu_erf_f((float)v[2 * j + 1]));
    *(h4v*)(out + i * 4) = o;
  }
}
int launch_geglu_erf_f16(tsd_ctx* ctx, const half_t* x, int64_t rows, int n_out, half_t* out) {
  if (n_out % 4) TSD_FAIL(TSD_E_SHAPE, "geglu: output width %d must be a multiple of 4", n_out);
  if (!ctx->launch()) return TSD_OK;
  const int64_t n = rows * (n_out / 4);
  hipLaunchKernelGGL(k_geglu_erf, GRID1D(n, 256), dim3(256), 0, ctx->stream, x, n, out);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
__global__ void k_add_const(float* __restrict__ dst, int64_t n, float c) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = __fadd_rn(dst[i], c);
}
int launch_add_const_f32(tsd_ctx* ctx, float* dst, int64_t n, float c) {
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_add_const, GRID1D(n, 256), dim3(256), 0, ctx->stream, dst, n, c);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
int launch_fill_uniform(tsd_ctx* ctx, float* dst, int64_t n, uint64_t seed, uint64_t tensor_id, float bound) {
  if (!ctx->launch()) return TSD_OK;
  const uint64_t base = seed * 0x9E3779B97F4A7C15ull + tensor_id * 0xBF58476D1CE4E5B9ull;
  hipLaunchKernelGGL(k_fill_uniform, GRID1D(n, 256), dim3(256), 0, ctx->stream, dst, n, base, bound);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- weight packing ------------------------------------------------------------------------------
// conv (O,I,k,k) fp32 -> fp16 [Opad][k*k][Ipad]: K index = tap*Ipad + i (tap-major so a 64-wide
// K chunk stays inside one tap and reads 128 contiguous NHWC bytes).  (Measured: making the taps the INNER K index
// for L2 reuse of the input rows is slower - the per-tile tap address update costs more than the locality buys.)
__global__ void k_pack_conv(const float* __restrict__ src, int O, int I, int kk, half_t* __restrict__ dst, int Ipad,
                            int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Ipad);
    const int tap = (int)((i / Ipad) % kk);
    const int o = (int)(i / ((int64_t)Ipad * kk));
    dst[i] = (o < O && ci < I) ? (half_t)src[((int64_t)o * I + ci) * kk + tap] : (half_t)0.f;
  }
}
int launch_pack_conv(tsd_ctx* ctx, const float* src, int O, int I, int k, half_t* dst, int Opad, int Ipad) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)Opad * k * k * Ipad;
  hipLaunchKernelGGL(k_pack_conv, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, O, I, k * k, dst, Ipad, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
// linear (N,K) fp32 -> fp16 [N][Kpad].  geglu_interleave: packed row 2q = row q ("a" half),
// 2q+1 = row N/2+q ("gate" half) so the GEMM epilogue sees (a,g) pairs in adjacent columns
// (`chunk(2,2)` + `out*gelu(gate)`, diffusion.mojo:138-141).
__global__ void k_pack_linear(const float* __restrict__ src, int N, int K, half_t* __restrict__ dst, int Kpad,
                              int inter, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const int r = (int)(i / Kpad);
    const int sr = inter ? ((r & 1) ? N / 2 + (r >> 1) : (r >> 1)) : r;
    dst[i] = (k < K) ? (half_t)src[(int64_t)sr * K + k] : (half_t)0.f;
  }
}
int launch_pack_linear(tsd_ctx* ctx, const float* src, int N, int K, half_t* dst, int Kpad, int geglu_interleave) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)N * Kpad;
  hipLaunchKernelGGL(k_pack_linear, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, N, K, dst, Kpad,
                     geglu_interleave, total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
__global__ void k_pack_bias(const float* __restrict__ src, int N, float* __restrict__ dst, int Npad, int inter) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= Npad) return;
  const int sr = inter ? ((r & 1) ? N / 2 + (r >> 1) : (r >> 1)) : r;
  dst[r] = (r < N) ? src[sr] : 0.f;
}
int launch_pack_bias(tsd_ctx* ctx, const float* src, int N, float* dst, int Npad, int geglu_interleave) {
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_pack_bias, dim3(ceil_div(Npad, 256)), dim3(256), 0, ctx->stream, src, N, dst, Npad,
                     geglu_interleave);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// ---- DDPM update + CFG combine (sampler.mojo:75-109, pipeline.mojo:117-119; App.D K9) ------------
// eps_hw > 0: eps / eps_u are the UNet output convolution's own layout [B][eps_hw][4] (x and noise stay CHW [B][4][eps_hw])
__global__ void k_ddpm_step(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ eps_u,
                            float cfg_scale, const float* __restrict__ noise, int64_t n, float sa, float sb,
                            float c_x0, float c_xt, float sigma, int eps_hw, int* __restrict__ nonfinite) {
  int nbad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t ie = i;
    if (eps_hw > 0) {
      const int64_t bc = i / eps_hw, pix = i - bc * eps_hw, b = bc >> 2;
      ie = (b * eps_hw + pix) * 4 + (bc & 3);
    }
    float e = eps[ie];
    if (eps_u) {
      const float u = eps_u[ie];
      e = (e - u) * cfg_scale + u;
    }
    const float xv = x[i];
    const float x0 = (xv - e * sb) / sa;
    float o = x0 * c_x0 + xv * c_xt;
    if (noise) o += noise[i] * sigma;
    nbad += nonfinite_f(o);  // a non-finite UNet output (fp16 overflow upstream) lands here every step
    x[i] = o;
  }
  nonfinite_report(nonfinite, nbad);
}
int launch_ddpm_step(tsd_ctx* ctx, float* latents, const float* eps, const float* eps_uncond, float cfg_scale,
                     const float* noise, int64_t n, float sa, float sb, float c_x0, float c_xt, float sigma, int eps_hw) {
  if (!ctx->launch()) return TSD_OK;
  ProfScope prof(ctx, KC_ELEMENTWISE);
  hipLaunchKernelGGL(k_ddpm_step, GRID1D(n, 256), dim3(256), 0, ctx->stream, latents, eps, eps_uncond, cfg_scale,
                     noise, n, sa, sb, c_x0, c_xt, sigma, eps_hw, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
__global__ void k_add_noise(float* __restrict__ x, const float* __restrict__ noise, int64_t n, float sa, float sb) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = x[i] * sa + noise[i] * sb;
}
int launch_add_noise(tsd_ctx* ctx, float* latents, const float* noise, int64_t n, float sa, float sb) {
  if (!ctx->launch()) return TSD_OK;
  hipLaunchKernelGGL(k_add_noise, GRID1D(n, 256), dim3(256), 0, ctx->stream, latents, noise, n, sa, sb);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
// `Encoder.metrics_evals` vae.mojo:118-129: moments NHWC fp32 [B][HW][ld] (mean ch 0..3, logvar 4..7)
__global__ void k_encoder_sample(const float* __restrict__ mom, int HW, int ld, const float* __restrict__ noise,
                                 float* __restrict__ out, int64_t total, int* __restrict__ nonfinite) {
  int nbad = 0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t pix = i % HW;
    const int c = (int)((i / HW) % 4);
    const int64_t b = i / ((int64_t)4 * HW);
    const float* mp = mom + (b * HW + pix) * ld;
    const float mean = mp[c];
    nbad += nonfinite_f(mean) + nonfinite_f(mp[4 + c]);  // the clamp below would swallow a NaN log-variance
    const float lv = fminf(fmaxf(mp[4 + c], -30.f), 20.f);
    const float sd = sqrtf(expf(lv));
    out[i] = (mean + noise[i] * sd) * 0.18215f;
  }
  nonfinite_report(nonfinite, nbad);
}
int launch_encoder_sample(tsd_ctx* ctx, const float* moments_nhwc, int B, int HW, int ld, const float* noise_chw,
                          float* latents_chw) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)B * 4 * HW;
  hipLaunchKernelGGL(k_encoder_sample, GRID1D(total, 256), dim3(256), 0, ctx->stream, moments_nhwc, HW, ld, noise_chw,
                     latents_chw, total, ctx->status);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}

// (batch,K,N) fp32 -> fp16 [batch][Npad][Kpad] (zero padded): the K-major "W" operand of `Matrix.matmul`
// (helpers/utils.mojo:1549-1569).
__global__ void k_transpose_to_f16(const float* __restrict__ src, int K, int N, half_t* __restrict__ dst, int Kpad,
                                   int Npad, int64_t total) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % Kpad);
    const int n = (int)((i / Kpad) % Npad);
    const int64_t b = i / ((int64_t)Kpad * Npad);
    dst[i] = (k < K && n < N) ? (half_t)src[(b * K + k) * N + n] : (half_t)0.f;
  }
}
int launch_transpose_f32_to_f16(tsd_ctx* ctx, const float* src, int batch, int K, int N, half_t* dst, int Kpad,
                                int Npad) {
  if (!ctx->launch()) return TSD_OK;
  const int64_t total = (int64_t)batch * Kpad * Npad;
  hipLaunchKernelGGL(k_transpose_to_f16, GRID1D(total, 256), dim3(256), 0, ctx->stream, src, K, N, dst, Kpad, Npad,
                     total);
  HIP_TRY(hipGetLastError());
  return TSD_OK;
}
