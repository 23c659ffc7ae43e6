// api_ops.cpp - op-level and block-level C-ABI entry points (host fp32 in/out, reference layouts).
// Each call uploads its operands, converts to the device layout (NHWC / K-major fp16), runs the same
// kernels the module path uses, converts back and synchronises - these are the parity/drop-in
// surface of the individual reference structs, not the measured path.
#include <math.h>

#include "graph.h"

namespace {
struct Dev {
  tsd_ctx* c;
  int err = TSD_OK;
  template <class T>
  T* buf(int64_t n) {
    T* d = arena_alloc<T>(c, n > 0 ? n : 1);
    if (!d && err == TSD_OK) { tsd_set_error("workspace arena exhausted"); err = TSD_E_ALLOC; }
    return d;
  }
  template <class T>
  T* in(const T* host, int64_t n) {
    T* d = buf<T>(n);
    if (d && c->launch() && err == TSD_OK) {
      hipError_t e = hipMemcpyAsync(d, host, (size_t)n * sizeof(T), hipMemcpyHostToDevice, c->stream);
      if (e != hipSuccess) { tsd_set_error("H2D copy failed: %s", hipGetErrorString(e)); err = TSD_E_HIP; }
    }
    return d;
  }
  template <class T>
  int out(T* host, const T* dev, int64_t n) {
    if (err != TSD_OK) return err;
    if (c->launch()) HIP_TRY(hipMemcpyAsync(host, dev, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, c->stream));
    return TSD_OK;
  }
  int out2d(float* host, const float* dev, int rows, int cols, int ld) {
    if (err != TSD_OK) return err;
    if (c->launch())
      HIP_TRY(hipMemcpy2DAsync(host, (size_t)cols * 4, dev, (size_t)ld * 4, (size_t)cols * 4, rows, hipMemcpyDeviceToHost,
                               c->stream));
    return TSD_OK;
  }
  int zero(void* p, size_t bytes) {
    if (c->launch()) HIP_TRY(hipMemsetAsync(p, 0, bytes, c->stream));
    return TSD_OK;
  }
  // pack host weights into the arena
  int conv(const float* w, const float* b, int O, int I, int k, int Opad, ConvW* o) {
    const int Ipad = round_up(I, 64);
    float* dw = in(w, (int64_t)O * I * k * k);
    half_t* pw = buf<half_t>((int64_t)Opad * k * k * Ipad);
    float* pb = buf<float>(Opad);
    if (err) return err;
    TSD_TRY(launch_pack_conv(c, dw, O, I, k, pw, Opad, Ipad));
    if (b) {
      float* db = in(b, O);
      if (err) return err;
      TSD_TRY(launch_pack_bias(c, db, O, pb, Opad, 0));
    } else TSD_TRY(zero(pb, (size_t)Opad * 4));
    o->w = pw; o->b = pb; o->I = I; o->O = O; o->k = k; o->Ipad = Ipad; o->Opad = Opad;
    return TSD_OK;
  }
  int lin(const float* w, const float* b, int N, int K, int interleave, LinW* o) {
    const int Kpad = round_up(K, 64), Npad = round_up(N, 4);
    float* dw = in(w, (int64_t)N * K);
    half_t* pw = buf<half_t>((int64_t)Npad * Kpad);
    if (err) return err;
    if (Npad != N) TSD_TRY(zero(pw + (int64_t)N * Kpad, (size_t)(Npad - N) * Kpad * 2));
    TSD_TRY(launch_pack_linear(c, dw, N, K, pw, Kpad, interleave));
    o->w = pw; o->N = N; o->K = K; o->Kpad = Kpad; o->b = nullptr;
    if (b) {
      float* db = in(b, N);
      float* pb = buf<float>(Npad);
      if (err) return err;
      TSD_TRY(launch_pack_bias(c, db, N, pb, Npad, interleave));
      o->b = pb;
    }
    return TSD_OK;
  }
};

template <class F>
int run_op(tsd_ctx* ctx, F&& fn) {
  if (!ctx) TSD_FAIL(TSD_E_ARG, "ctx is NULL");
  HIP_TRY(hipSetDevice(ctx->device));
  int r = run_planned(ctx, fn);
  if (r != TSD_OK) {
    hipStreamSynchronize(ctx->stream);
    return r;
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return ctx_check_status(ctx);  // split-K hand-off time-outs, non-finite outputs (fp16 overflow): never silent
}
}  // namespace

#define NOTNULL(p) \
  if (!(p)) TSD_FAIL(TSD_E_ARG, "%s: argument '%s' is NULL", __func__, #p)

extern "C" int tsd_conv2d_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, const float* w, const float* bias,
                              int I, int O, int k, int pad_h, int pad_w, int stride_h, int stride_w, float* y) {
  NOTNULL(x); NOTNULL(w); NOTNULL(y);
  if (C <= 0 || H <= 0 || W <= 0 || O <= 0 || I <= 0 || I > C) TSD_FAIL(TSD_E_SHAPE, "conv2d: bad dims C=%d I=%d O=%d", C, I, O);
  if (k != 1 && k != 3) TSD_FAIL(TSD_E_SHAPE, "conv2d: kernel size %d unsupported (1 or 3)", k);
  if (pad_h != pad_w || stride_h != stride_w || stride_h < 1 || pad_h < 0)
    TSD_FAIL(TSD_E_SHAPE, "conv2d: only square padding/stride are on the path");
  if (k == 1 && (pad_h != 0 || stride_h != 1)) TSD_FAIL(TSD_E_SHAPE, "conv2d: 1x1 conv with padding/stride unsupported");
  const int Ho = (H + 2 * pad_h - k) / stride_h + 1, Wo = (W + 2 * pad_w - k) / stride_w + 1;
  if (Ho <= 0 || Wo <= 0) TSD_FAIL(TSD_E_SHAPE, "conv2d: empty output");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    const int Opad = round_up(O, 4);
    ConvW cw;
    TSD_TRY(d.conv(w, bias, O, I, k, Opad, &cw));
    float* dx = d.in(x, (int64_t)C * H * W);
    Act xa = act_alloc(ctx, 1, H, W, cw.Ipad);
    float* y32 = d.buf<float>((int64_t)Ho * Wo * Opad);
    float* ychw = d.buf<float>((int64_t)O * Ho * Wo);
    if (d.err || !xa.p) return d.err ? d.err : TSD_E_ALLOC;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, C, H, W, I, 1.f, xa.p, cw.Ipad));
    if (k == 3) TSD_TRY(g_conv3x3(ctx, xa, cw, stride_h, pad_h, pad_h, 0, nullptr, 0, nullptr, 0, true, y32, Opad));
    else {
      GemmArgs g;
      g.A0 = xa.p; g.lda0 = xa.ld; g.Wt = cw.w; g.ldw = cw.Ipad; g.M = H * W; g.N = Opad; g.K = cw.Ipad;
      g.epi = EPI_BIAS_N | EPI_OUT_F32; g.bias = cw.b; g.C = y32; g.ldc = Opad;
      TSD_TRY(launch_gemm(ctx, g));
    }
    TSD_TRY(launch_nhwc_f32_to_chw_f32(ctx, y32, 1, O, Ho, Wo, Opad, ychw));
    return d.out(y, ychw, (int64_t)O * Ho * Wo);
  });
}

extern "C" int tsd_pad_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int top, int bottom, int left, int right,
                           float* y) {
  NOTNULL(x); NOTNULL(y);
  if (C <= 0 || H <= 0 || W <= 0 || top < 0 || bottom < 0 || left < 0 || right < 0) TSD_FAIL(TSD_E_SHAPE, "pad: bad dims");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    const int64_t n_out = (int64_t)C * (H + top + bottom) * (W + left + right);
    float* dx = d.in(x, (int64_t)C * H * W);
    float* dy = d.buf<float>(n_out);
    if (d.err) return d.err;
    TSD_TRY(launch_pad_f32(ctx, dx, C, H, W, top, bottom, left, right, dy));
    return d.out(y, dy, n_out);
  });
}

extern "C" int tsd_groupnorm_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int groups, int num_channels,
                                 float eps, float gamma, float* y) {
  NOTNULL(x); NOTNULL(y);
  // reference checks, helpers/utils.mojo:1847-1853 ("Returning null matrix")
  if (num_channels > C) TSD_FAIL(TSD_E_SHAPE, "groupnorm: num_channels %d exceeds input channels %d", num_channels, C);
  if (groups <= 0 || num_channels % groups) TSD_FAIL(TSD_E_SHAPE, "groupnorm: %d channels not divisible by %d groups", num_channels, groups);
  if (num_channels % 8) TSD_FAIL(TSD_E_SHAPE, "groupnorm: num_channels %d must be a multiple of 8 on the device path", num_channels);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    const int Cn = num_channels;
    float* dx = d.in(x, (int64_t)C * H * W);
    half_t* x16 = d.buf<half_t>((int64_t)H * W * Cn);
    half_t* y16 = d.buf<half_t>((int64_t)H * W * Cn);
    float* dy = d.buf<float>((int64_t)Cn * H * W);
    if (d.err) return d.err;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, C, H, W, Cn, 1.f, x16, Cn));
    NormSrc s; s.x0 = x16; s.ld0 = Cn; s.C0 = Cn;
    TSD_TRY(launch_groupnorm(ctx, s, 1, H * W, Cn, groups, eps, gamma, 0, y16, Cn));
    TSD_TRY(launch_nhwc_f16_to_chw_f32(ctx, y16, 1, Cn, H, W, Cn, dy));
    return d.out(y, dy, (int64_t)Cn * H * W);
  });
}

extern "C" int tsd_layernorm_f32(tsd_ctx* ctx, const float* x, int M, int C, float eps, float* y) {
  NOTNULL(x); NOTNULL(y);
  if (M <= 0 || C <= 0 || C % 8) TSD_FAIL(TSD_E_SHAPE, "layernorm: C=%d must be a positive multiple of 8", C);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dx = d.in(x, (int64_t)M * C);
    half_t* x16 = d.buf<half_t>((int64_t)M * C);
    half_t* y16 = d.buf<half_t>((int64_t)M * C);
    float* dy = d.buf<float>((int64_t)M * C);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, dx, M, C, x16, C, M));
    TSD_TRY(launch_layernorm(ctx, x16, M, C, C, eps, y16, C));
    TSD_TRY(launch_f16_to_f32_rows(ctx, y16, M, C, C, dy));
    return d.out(y, dy, (int64_t)M * C);
  });
}

extern "C" int tsd_groupnorm_affine_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int groups, float eps,
                                        const float* weight, const float* bias, int silu, float* y) {
  NOTNULL(x); NOTNULL(y);
  if (groups <= 0 || C % groups) TSD_FAIL(TSD_E_SHAPE, "groupnorm: %d channels not divisible by %d groups", C, groups);
  if (C % 8) TSD_FAIL(TSD_E_SHAPE, "groupnorm: C=%d must be a multiple of 8 on the device path", C);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dx = d.in(x, (int64_t)C * H * W);
    float* dw = weight ? d.in(weight, C) : nullptr;
    float* db = bias ? d.in(bias, C) : nullptr;
    half_t* x16 = d.buf<half_t>((int64_t)H * W * C);
    half_t* y16 = d.buf<half_t>((int64_t)H * W * C);
    float* dy = d.buf<float>((int64_t)C * H * W);
    if (d.err) return d.err;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, C, H, W, C, 1.f, x16, C));
    NormSrc s; s.x0 = x16; s.ld0 = C; s.C0 = C;
    NormAffine a; a.w = dw; a.b = db; a.torch_rstd = 1;
    TSD_TRY(launch_groupnorm(ctx, s, 1, H * W, C, groups, eps, 1.f, silu ? 1 : 0, y16, C, nullptr, 0, &a));
    TSD_TRY(launch_nhwc_f16_to_chw_f32(ctx, y16, 1, C, H, W, C, dy));
    return d.out(y, dy, (int64_t)C * H * W);
  });
}

extern "C" int tsd_layernorm_affine_f32(tsd_ctx* ctx, const float* x, int M, int C, float eps, const float* weight,
                                        const float* bias, float* y) {
  NOTNULL(x); NOTNULL(y);
  if (M <= 0 || C <= 0 || C % 8) TSD_FAIL(TSD_E_SHAPE, "layernorm: C=%d must be a positive multiple of 8", C);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dx = d.in(x, (int64_t)M * C);
    float* dw = weight ? d.in(weight, C) : nullptr;
    float* db = bias ? d.in(bias, C) : nullptr;
    half_t* x16 = d.buf<half_t>((int64_t)M * C);
    half_t* y16 = d.buf<half_t>((int64_t)M * C);
    float* dy = d.buf<float>((int64_t)M * C);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, dx, M, C, x16, C, M));
    NormAffine a; a.w = dw; a.b = db; a.torch_rstd = 1;
    TSD_TRY(launch_layernorm(ctx, x16, M, C, C, eps, y16, C, &a));
    TSD_TRY(launch_f16_to_f32_rows(ctx, y16, M, C, C, dy));
    return d.out(y, dy, (int64_t)M * C);
  });
}

static int unary_op(tsd_ctx* ctx, int op, const float* x, int64_t n, float* y) {
  if (!x || !y) TSD_FAIL(TSD_E_ARG, "unary op: NULL argument");
  if (n <= 0) TSD_FAIL(TSD_E_SHAPE, "unary op: n=%lld", (long long)n);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dx = d.in(x, n);
    float* dy = d.buf<float>(n);
    if (d.err) return d.err;
    TSD_TRY(launch_unary_f32(ctx, op, dx, n, dy));
    return d.out(y, dy, n);
  });
}
extern "C" int tsd_silu_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y) { return unary_op(ctx, 0, x, n, y); }
extern "C" int tsd_gelu_tanh_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y) { return unary_op(ctx, 1, x, n, y); }
extern "C" int tsd_rescale_images_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y) { return unary_op(ctx, 2, x, n, y); }

extern "C" int tsd_linear_f32(tsd_ctx* ctx, const float* x, int M, int K, const float* w, const float* bias, int N,
                              float* y) {
  NOTNULL(x); NOTNULL(w); NOTNULL(y);
  if (M <= 0 || K <= 0 || N <= 0) TSD_FAIL(TSD_E_SHAPE, "linear: bad dims M=%d K=%d N=%d", M, K, N);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    LinW lw;
    TSD_TRY(d.lin(w, bias, N, K, 0, &lw));
    const int Npad = round_up(N, 4);
    float* dx = d.in(x, (int64_t)M * K);
    half_t* x16 = d.buf<half_t>((int64_t)M * lw.Kpad);
    float* y32 = d.buf<float>((int64_t)M * Npad);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, dx, M, K, x16, lw.Kpad, M));
    GemmArgs g;
    g.A0 = x16; g.lda0 = lw.Kpad; g.Wt = lw.w; g.ldw = lw.Kpad; g.M = M; g.N = Npad; g.K = lw.Kpad;
    g.epi = EPI_OUT_F32 | (lw.b ? EPI_BIAS_N : 0); g.bias = lw.b; g.C = y32; g.ldc = Npad;
    TSD_TRY(launch_gemm(ctx, g));
    return d.out2d(y, y32, M, N, Npad);
  });
}

extern "C" int tsd_matmul_f32(tsd_ctx* ctx, const float* a, const float* bmat, int batch, int b_batch, int M, int K,
                              int N, float* c) {
  NOTNULL(a); NOTNULL(bmat); NOTNULL(c);
  if (batch <= 0 || M <= 0 || K <= 0 || N <= 0 || (b_batch != 1 && b_batch != batch))
    TSD_FAIL(TSD_E_SHAPE, "matmul: bad dims batch=%d b_batch=%d M=%d K=%d N=%d", batch, b_batch, M, K, N);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    const int Kpad = round_up(K, 64), Npad = round_up(N, 4);
    float* da = d.in(a, (int64_t)batch * M * K);
    float* db = d.in(bmat, (int64_t)b_batch * K * N);
    half_t* a16 = d.buf<half_t>((int64_t)batch * M * Kpad);
    half_t* b16 = d.buf<half_t>((int64_t)b_batch * Npad * Kpad);
    float* c32 = d.buf<float>((int64_t)batch * M * Npad);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, da, (int64_t)batch * M, K, a16, Kpad, (int64_t)batch * M));
    TSD_TRY(launch_transpose_f32_to_f16(ctx, db, b_batch, K, N, b16, Kpad, Npad));
    GemmArgs g;
    g.A0 = a16; g.lda0 = Kpad; g.sA = (int64_t)M * Kpad;
    g.Wt = b16; g.ldw = Kpad; g.sW = b_batch == 1 ? 0 : (int64_t)Npad * Kpad;
    g.M = M; g.N = Npad; g.K = Kpad; g.batch = batch;
    g.epi = EPI_OUT_F32; g.C = c32; g.ldc = Npad; g.sC = (int64_t)M * Npad;
    TSD_TRY(launch_gemm(ctx, g));
    return d.out2d(c, c32, batch * M, N, Npad);
  });
}

extern "C" int tsd_upsample_nearest2x_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, float* y) {
  NOTNULL(x); NOTNULL(y);
  if (C <= 0 || H <= 0 || W <= 0) TSD_FAIL(TSD_E_SHAPE, "upsample: bad dims");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    const int64_t n = (int64_t)C * H * W;
    float* dx = d.in(x, n);
    float* dy = d.buf<float>(4 * n);
    if (d.err) return d.err;
    TSD_TRY(launch_upsample_f32(ctx, dx, C, H, W, dy));
    return d.out(y, dy, 4 * n);
  });
}

extern "C" int tsd_softmax_lastdim_f32(tsd_ctx* ctx, const float* x, int64_t rows, int cols, float* y) {
  NOTNULL(x); NOTNULL(y);
  if (rows <= 0 || cols <= 0) TSD_FAIL(TSD_E_SHAPE, "softmax: bad dims");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dx = d.in(x, rows * cols);
    float* dy = d.buf<float>(rows * cols);
    if (d.err) return d.err;
    TSD_TRY(launch_softmax_rows_f32(ctx, dx, rows, cols, dy));
    return d.out(y, dy, rows * cols);
  });
}

extern "C" int tsd_time_embedding_f32(tsd_ctx* ctx, float t, float* out320) {
  NOTNULL(out320);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    float* dy = d.buf<float>(320);
    if (d.err) return d.err;
    TSD_TRY(launch_time_embedding(ctx, nullptr, t, 1, dy));
    return d.out(out320, dy, 320);
  });
}

// shared by self/cross attention: x16 [T][Dpad] -> out
static int attention_tail(tsd_ctx* ctx, Dev& d, const AttnArgs& fa, const LinW& wo, int T, int D, float* y) {
  TSD_TRY(g_attn_core(ctx, fa));
  const int Npad = round_up(D, 4);
  float* y32 = d.buf<float>((int64_t)T * Npad);
  if (d.err) return d.err;
  GemmArgs g;
  g.A0 = fa.O; g.lda0 = fa.ldo; g.Wt = wo.w; g.ldw = wo.Kpad; g.M = T; g.N = Npad; g.K = wo.Kpad;
  g.epi = EPI_OUT_F32 | (wo.b ? EPI_BIAS_N : 0); g.bias = wo.b; g.C = y32; g.ldc = Npad;
  TSD_TRY(launch_gemm(ctx, g));
  return d.out2d(y, y32, T, D, Npad);
}

extern "C" int tsd_self_attention_f32(tsd_ctx* ctx, const float* x, int T, int D, int heads, const float* w_in,
                                      const float* b_in, const float* w_out, const float* b_out, int causal,
                                      float* y) {
  NOTNULL(x); NOTNULL(w_in); NOTNULL(w_out); NOTNULL(y);
  if (causal) TSD_FAIL(TSD_E_SHAPE, "self_attention: causal masking is CLIP-only and not on the device path");
  if (T <= 0 || D <= 0 || heads <= 0 || D % heads || D % 64 || T % 8)
    TSD_FAIL(TSD_E_SHAPE, "self_attention: T=%d D=%d heads=%d unsupported (D %% 64, T %% 8)", T, D, heads);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    LinW wi, wo;
    TSD_TRY(d.lin(w_in, b_in, 3 * D, D, 0, &wi));
    TSD_TRY(d.lin(w_out, b_out, D, D, 0, &wo));
    float* dx = d.in(x, (int64_t)T * D);
    half_t* x16 = d.buf<half_t>((int64_t)T * D);
    half_t* qk = d.buf<half_t>((int64_t)T * 2 * D);
    half_t* vt = d.buf<half_t>((int64_t)D * T);
    half_t* ao = d.buf<half_t>((int64_t)T * D);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, dx, T, D, x16, D, T));
    TSD_TRY(g_qkv_proj(ctx, x16, 1, T, D, wi, qk, vt, T));
    AttnArgs fa;
    fa.Q = qk; fa.ldq = 2 * D; fa.K = qk + D; fa.ldk = 2 * D; fa.Vt = vt; fa.ldvt = T; fa.O = ao; fa.ldo = D;
    fa.B = 1; fa.H = heads; fa.d = D / heads; fa.Sq = T; fa.Sk = T; fa.scale = 1.f / sqrtf((float)(D / heads));
    return attention_tail(ctx, d, fa, wo, T, D, y);
  });
}

extern "C" int tsd_cross_attention_f32(tsd_ctx* ctx, const float* x, int Tq, int D, const float* context, int Tk,
                                       int Dc, int heads, const float* wq, const float* bq, const float* wk,
                                       const float* bk, const float* wv, const float* bv, const float* wo,
                                       const float* bo, float* y) {
  NOTNULL(x); NOTNULL(context); NOTNULL(wq); NOTNULL(wk); NOTNULL(wv); NOTNULL(wo); NOTNULL(y);
  if (Tq <= 0 || Tk <= 0 || D <= 0 || Dc <= 0 || heads <= 0 || D % heads || D % 64)
    TSD_FAIL(TSD_E_SHAPE, "cross_attention: Tq=%d Tk=%d D=%d Dc=%d heads=%d unsupported", Tq, Tk, D, Dc, heads);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    LinW lq, lk, lv, lo;
    TSD_TRY(d.lin(wq, bq, D, D, 0, &lq));
    TSD_TRY(d.lin(wk, bk, D, Dc, 0, &lk));
    TSD_TRY(d.lin(wv, bv, D, Dc, 0, &lv));
    TSD_TRY(d.lin(wo, bo, D, D, 0, &lo));
    const int Tp = round_up(Tk, 8);
    float* dx = d.in(x, (int64_t)Tq * D);
    float* dc = d.in(context, (int64_t)Tk * Dc);
    half_t* x16 = d.buf<half_t>((int64_t)Tq * D);
    half_t* c16 = d.buf<half_t>((int64_t)Tp * lk.Kpad);
    half_t* q = d.buf<half_t>((int64_t)Tq * D);
    half_t* kc = d.buf<half_t>((int64_t)Tp * D);
    half_t* vtc = d.buf<half_t>((int64_t)D * Tp);
    half_t* ao = d.buf<half_t>((int64_t)Tq * D);
    if (d.err) return d.err;
    TSD_TRY(launch_f32_to_f16_rows(ctx, dx, Tq, D, x16, D, Tq));
    TSD_TRY(launch_f32_to_f16_rows(ctx, dc, Tk, Dc, c16, lk.Kpad, Tp));
    CatSrc a; a.p0 = x16; a.ld0 = D; a.C0 = D;
    TSD_TRY(g_linear(ctx, a, Tq, lq.w, lq.Kpad, D, D, lq.b, nullptr, 0, 0, q, D));
    CatSrc ac; ac.p0 = c16; ac.ld0 = lk.Kpad; ac.C0 = lk.Kpad;
    TSD_TRY(g_linear(ctx, ac, Tp, lk.w, lk.Kpad, D, lk.Kpad, lk.b, nullptr, 0, 0, kc, D));
    {
      GemmArgs g;  // V^T = W_v . ctx^T (+ b_v per row)
      g.A0 = lv.w; g.lda0 = lv.Kpad; g.Wt = c16; g.ldw = lv.Kpad; g.M = D; g.N = Tp; g.K = lv.Kpad;
      if (lv.b) { g.epi = EPI_BIAS_M; g.bias = lv.b; }
      g.C = vtc; g.ldc = Tp;
      TSD_TRY(launch_gemm(ctx, g));
    }
    AttnArgs fa;
    fa.Q = q; fa.ldq = D; fa.K = kc; fa.ldk = D; fa.Vt = vtc; fa.ldvt = Tp; fa.O = ao; fa.ldo = D;
    fa.B = 1; fa.H = heads; fa.d = D / heads; fa.Sq = Tq; fa.Sk = Tk; fa.scale = 1.f / sqrtf((float)(D / heads));
    return attention_tail(ctx, d, fa, lo, Tq, D, y);
  });
}

// ---- block level ------------------------------------------------------------------------------------
extern "C" int tsd_time_embedding_mlp_f32(tsd_ctx* ctx, const float* t320, const float* w1, const float* b1,
                                          const float* w2, const float* b2, float* out1280) {
  NOTNULL(t320); NOTNULL(w1); NOTNULL(b1); NOTNULL(w2); NOTNULL(b2); NOTNULL(out1280);
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    LinW l1, l2;
    TSD_TRY(d.lin(w1, b1, 1280, 320, 0, &l1));
    TSD_TRY(d.lin(w2, b2, 1280, 1280, 0, &l2));
    float* dt = d.in(t320, 320);
    float* h = d.buf<float>(1280);
    float* o = d.buf<float>(1280);
    if (d.err) return d.err;
    TSD_TRY(launch_small_linear(ctx, dt, 1, 320, 320, l1.w, l1.Kpad, l1.b, 1280, 0, h, 1280));
    TSD_TRY(launch_small_linear(ctx, h, 1, 1280, 1280, l2.w, l2.Kpad, l2.b, 1280, 1, o, 1280));
    return d.out(out1280, o, 1280);
  });
}

static int res_block_common(tsd_ctx* ctx, const float* x, int Cx, int H, int W, const float* time, int cin, int cout,
                            int groups, const float* conv1_w, const float* conv1_b, const float* lin_w,
                            const float* lin_b, const float* conv2_w, const float* conv2_b, const float* skip_w,
                            const float* skip_b, float* y) {
  if (cin % 64 || cout % 64 || Cx < cin || H <= 0 || W <= 0)
    TSD_FAIL(TSD_E_SHAPE, "residual block: cin=%d cout=%d Cx=%d unsupported (multiples of 64)", cin, cout, Cx);
  if (cin != cout && !skip_w) TSD_FAIL(TSD_E_ARG, "residual block: skip conv weights required when cin != cout");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    ResW rw;
    rw.cin = cin; rw.cout = cout; rw.groups = groups; rw.has_skip = cin != cout;
    TSD_TRY(d.conv(conv1_w, conv1_b, cout, cin, 3, cout, &rw.conv1));
    TSD_TRY(d.conv(conv2_w, conv2_b, cout, cout, 3, cout, &rw.conv2));
    if (rw.has_skip) TSD_TRY(d.conv(skip_w, skip_b, cout, cin, 1, cout, &rw.skip));
    float* tvec = nullptr;
    if (time) {
      LinW lt;
      TSD_TRY(d.lin(lin_w, lin_b, cout, 1280, 0, &lt));
      float* dt = d.in(time, 1280);
      tvec = d.buf<float>(cout);
      if (d.err) return d.err;
      TSD_TRY(launch_small_linear(ctx, dt, 1, 1280, 1280, lt.w, lt.Kpad, lt.b, cout, 1, tvec, cout));  // diffusion.mojo:61-62
    }
    float* dx = d.in(x, (int64_t)Cx * H * W);
    Act xa = act_alloc(ctx, 1, H, W, cin);
    Act out = act_alloc(ctx, 1, H, W, cout);
    float* dy = d.buf<float>((int64_t)cout * H * W);
    if (d.err || !xa.p || !out.p) return d.err ? d.err : TSD_E_ALLOC;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, Cx, H, W, cin, 1.f, xa.p, cin));
    TSD_TRY(g_resblock(ctx, cat1(xa), 1, H, W, 0, rw, tvec, cout, out));
    TSD_TRY(launch_nhwc_f16_to_chw_f32(ctx, out.p, 1, cout, H, W, cout, dy));
    return d.out(y, dy, (int64_t)cout * H * W);
  });
}

extern "C" int tsd_unet_residual_block_f32(tsd_ctx* ctx, const float* x, int Cx, int H, int W, const float* time,
                                           int cin, int cout, const float* conv1_w, const float* conv1_b,
                                           const float* lin_w, const float* lin_b, const float* conv2_w,
                                           const float* conv2_b, const float* skip_w, const float* skip_b, float* y) {
  NOTNULL(x); NOTNULL(time); NOTNULL(conv1_w); NOTNULL(conv1_b); NOTNULL(lin_w); NOTNULL(lin_b); NOTNULL(conv2_w);
  NOTNULL(conv2_b); NOTNULL(y);
  return res_block_common(ctx, x, Cx, H, W, time, cin, cout, 32, conv1_w, conv1_b, lin_w, lin_b, conv2_w, conv2_b,
                          skip_w, skip_b, y);
}

extern "C" int tsd_vae_res_block_f32(tsd_ctx* ctx, const float* x, int H, int W, int cin, int cout,
                                     const float* conv1_w, const float* conv1_b, const float* conv2_w,
                                     const float* conv2_b, const float* skip_w, const float* skip_b, float* y) {
  NOTNULL(x); NOTNULL(conv1_w); NOTNULL(conv1_b); NOTNULL(conv2_w); NOTNULL(conv2_b); NOTNULL(y);
  return res_block_common(ctx, x, cin, H, W, nullptr, cin, cout, 16, conv1_w, conv1_b, nullptr, nullptr, conv2_w,
                          conv2_b, skip_w, skip_b, y);
}

extern "C" int tsd_unet_attention_block_f32(tsd_ctx* ctx, const float* x, int n_head, int n_embed, int H, int W,
                                            const float* context, int Tk, int Dc, const float* const* w, int nw,
                                            float* y) {
  NOTNULL(x); NOTNULL(context); NOTNULL(w); NOTNULL(y);
  if (nw != 16) TSD_FAIL(TSD_E_ARG, "attention block: expected 16 weight pointers, got %d", nw);
  for (int i = 0; i < 16; i++) if (!w[i]) TSD_FAIL(TSD_E_ARG, "attention block: weight pointer %d is NULL", i);
  const int C = n_head * n_embed;
  if (C <= 0 || C % 64 || H <= 0 || W <= 0 || Tk <= 0 || Dc <= 0) TSD_FAIL(TSD_E_SHAPE, "attention block: bad dims");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    AttnW aw;
    aw.n_head = n_head; aw.n_embed = n_embed; aw.C = C; aw.d_ctx = Dc;
    TSD_TRY(d.conv(w[0], w[1], C, C, 1, C, &aw.conv_in));
    TSD_TRY(d.lin(w[2], nullptr, 3 * C, C, 0, &aw.sa_in));      // in_bias=False, diffusion.mojo:92
    TSD_TRY(d.lin(w[3], w[4], C, C, 0, &aw.sa_out));
    TSD_TRY(d.lin(w[5], nullptr, C, C, 0, &aw.ca_q));           // in_bias=False, diffusion.mojo:94
    TSD_TRY(d.lin(w[6], nullptr, C, Dc, 0, &aw.ca_k));
    TSD_TRY(d.lin(w[7], nullptr, C, Dc, 0, &aw.ca_v));
    TSD_TRY(d.lin(w[8], w[9], C, C, 0, &aw.ca_out));
    TSD_TRY(d.lin(w[10], w[11], 8 * C, C, 1, &aw.geglu1));
    TSD_TRY(d.lin(w[12], w[13], C, 4 * C, 0, &aw.geglu2));
    TSD_TRY(d.conv(w[14], w[15], C, C, 1, C, &aw.conv_out));
    const int Tp = round_up(Tk, 8);
    float* dx = d.in(x, (int64_t)C * H * W);
    float* dc = d.in(context, (int64_t)Tk * Dc);
    Act xa = act_alloc(ctx, 1, H, W, C);
    Act out = act_alloc(ctx, 1, H, W, C);
    half_t* c16 = d.buf<half_t>((int64_t)Tp * aw.ca_k.Kpad);
    float* dy = d.buf<float>((int64_t)C * H * W);
    if (d.err || !xa.p || !out.p) return d.err ? d.err : TSD_E_ALLOC;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, C, H, W, C, 1.f, xa.p, C));
    TSD_TRY(launch_f32_to_f16_rows(ctx, dc, Tk, Dc, c16, aw.ca_k.Kpad, Tp));
    TSD_TRY(g_unet_attn(ctx, xa, aw, c16, Tk, Tp, out));
    TSD_TRY(launch_nhwc_f16_to_chw_f32(ctx, out.p, 1, C, H, W, C, dy));
    return d.out(y, dy, (int64_t)C * H * W);
  });
}

extern "C" int tsd_vae_attention_block_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, const float* w_in,
                                           const float* b_in, const float* w_out, const float* b_out, float* y) {
  NOTNULL(x); NOTNULL(w_in); NOTNULL(b_in); NOTNULL(w_out); NOTNULL(b_out); NOTNULL(y);
  if (C <= 0 || C % 64 || H <= 0 || W <= 0) TSD_FAIL(TSD_E_SHAPE, "vae attention block: bad dims");
  return run_op(ctx, [&]() -> int {
    Dev d{ctx};
    VaeAttnW vw;
    vw.C = C;
    TSD_TRY(d.lin(w_in, b_in, 3 * C, C, 0, &vw.in_proj));
    TSD_TRY(d.lin(w_out, b_out, C, C, 0, &vw.out_proj));
    float* dx = d.in(x, (int64_t)C * H * W);
    Act xa = act_alloc(ctx, 1, H, W, C);
    Act out = act_alloc(ctx, 1, H, W, C);
    float* dy = d.buf<float>((int64_t)C * H * W);
    if (d.err || !xa.p || !out.p) return d.err ? d.err : TSD_E_ALLOC;
    TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, dx, 1, C, H, W, C, 1.f, xa.p, C));
    TSD_TRY(g_vae_attn(ctx, xa, vw, out));
    TSD_TRY(launch_nhwc_f16_to_chw_f32(ctx, out.p, 1, C, H, W, C, dy));
    return d.out(y, dy, (int64_t)C * H * W);
  });
}
