// graph.cpp - block and module graphs of the hot path (host code; every op is a kernel launch on
// the context stream, activations are NHWC fp16 views into the workspace arena).
#include <math.h>

#include "graph.h"

#define CHECK_ALLOC(p) \
  if (!(p)) TSD_FAIL(TSD_E_ALLOC, "workspace arena exhausted (%s:%d)", __FILE__, __LINE__)

static int zero_async(tsd_ctx* ctx, void* p, size_t bytes) {
  if (!ctx->launch()) return TSD_OK;
  HIP_TRY(hipMemsetAsync(p, 0, bytes, ctx->stream));
  return TSD_OK;
}

Act act_alloc_gn(tsd_ctx* ctx, int B, int H, int W, int C, int groups) {
  Act a = act_alloc(ctx, B, H, W, C);
  if (a.p && groups > 0 && C % groups == 0) {
    a.gn_buf = arena_alloc<float>(ctx, (int64_t)B * ceil_div(H * W, 32) * groups * 2);
    a.gn_groups = a.gn_buf ? groups : 0;
  }
  return a;
}

// ask the GEMM/conv that writes `dst` to emit the statistics of the GroupNorm that will read it (EPI_GNSTATS); a tile
// geometry that cannot do it leaves dst->gn_part NULL and the consumer runs its own statistics pass.
static void gn_emit(const tsd_ctx* ctx, GemmArgs& g, Act* dst, int rows_per_sample) {
  if (!dst || !dst->gn_buf || dst->gn_groups <= 0 || g.N != dst->C || (g.epi & (EPI_OUT_F32 | EPI_GEGLU))) return;
  const int ns = gemm_gnstats_slabs(ctx, g.M, g.N, g.K, g.batch, g.conv, rows_per_sample, dst->gn_groups);
  // up to 128 slabs the apply blocks (or k_gn_finalize) reduce them directly; beyond 256 (the VAE's 128^2 ... 512^2 images)
  // launch_groupnorm pre-reduces them to 64 chunks per sample (k_gn_prereduce) - either way no statistics pass over the tensor
  if (ns <= 0 || ns > ceil_div(rows_per_sample, 32) || (ns > 128 && ns <= 256) || (ns > 256 && dst->gn_groups > 256)) return;  // the 256-group limit is k_gn_prereduce's
  g.epi |= EPI_GNSTATS;
  g.gn_part = dst->gn_buf; g.gn_groups = dst->gn_groups; g.gn_rows_per_sample = rows_per_sample; g.gn_nslab = ns;
  dst->gn_part = dst->gn_buf; dst->gn_nslab = ns;
}

int g_conv3x3(tsd_ctx* ctx, const Act& x, const ConvW& w, int stride, int pad, int pad_br, int ups,
              const float* rowvec, int rowvec_ld, const Act* res, int res_ups, bool out_f32, void* y, int ldy, Act* stat,
              const CatSrc* skip_x, const ConvW* skip_w) {
  if (w.k != 3) TSD_FAIL(TSD_E_SHAPE, "g_conv3x3: kernel size %d", w.k);
  if (x.ld < w.Ipad) TSD_FAIL(TSD_E_SHAPE, "g_conv3x3: input pitch %d < padded Cin %d", x.ld, w.Ipad);
  const int Hin = ups ? 2 * x.H : x.H, Win = ups ? 2 * x.W : x.W;
  // Ho = floor((H + pad + pad_br - k)/s) + 1 (helpers/utils.mojo:1752-1758); pad = top/left, pad_br =
  // bottom/right (equal for Conv2D's symmetric padding; (0,1) for the encoder's two_stride_pad, vae.mojo:115-116).
  const int pad_total = pad + pad_br;
  const int Ho = (Hin + pad_total - 3) / stride + 1, Wo = (Win + pad_total - 3) / stride + 1;
  GemmArgs g;
  g.A0 = x.p; g.lda0 = x.ld; g.conv = 1; g.Hs = x.H; g.Ws = x.W; g.Ho = Ho; g.Wo = Wo; g.Cin = w.Ipad;
  g.stride = stride; g.pad = pad; g.ups = ups;
  g.Wt = w.w; g.ldw = 9 * w.Ipad;
  if (w.w_tm) { g.Wt = w.w_tm; g.ldw = 64; g.w_kts = w.Opad * 128; }  // K-tile-major copy (weight-heavy layers)
  g.M = x.B * Ho * Wo; g.N = w.Opad; g.K = 9 * w.Ipad;
  g.epi = EPI_BIAS_N; g.bias = w.b;
  if (rowvec) { g.epi |= EPI_ROWVEC; g.rowvec = rowvec; g.rowvec_ld = rowvec_ld; g.rows_per_batch = Ho * Wo; }
  if (res) { g.epi |= EPI_RESIDUAL | (res_ups ? EPI_RES_UPS : 0); g.R = res->p; g.ldr = res->ld; }
  if (out_f32) g.epi |= EPI_OUT_F32;
  if (skip_x && skip_w) {
    // y += conv1x1(skip_x) + skip bias: K runs on past the nine taps over the skip tensor's channels (kernels_gemm.hip "taps" 9 / 10);
    // the 1x1 bias rides in the per-sample row vector slot with pitch 0
    if (rowvec || skip_w->k != 1 || skip_w->Opad != w.Opad) TSD_FAIL(TSD_E_ARG, "g_conv3x3: fused skip does not fit");
    const int ck = skip_w->Ipad;
    g.A1 = skip_x->p0; g.lda1 = skip_x->ld0; g.Cin1 = (skip_x->p1 && ck > skip_x->C0) ? skip_x->C0 : ck;
    if (g.Cin1 < ck) { g.A2 = skip_x->p1; g.lda2 = skip_x->ld1; g.Cin2 = ck - g.Cin1; }
    g.Wt1 = skip_w->w; g.ldw1 = ck;
    g.K += ck;
    if (skip_w->b) { g.epi |= EPI_ROWVEC; g.rowvec = skip_w->b; g.rowvec_ld = 0; g.rows_per_batch = Ho * Wo; }
  }
  g.C = y; g.ldc = ldy;
  g.rows_per_sample_hint = Ho * Wo;
  gn_emit(ctx, g, stat, Ho * Wo);
  return launch_gemm(ctx, g);
}

int g_linear(tsd_ctx* ctx, const CatSrc& a, int64_t M, const half_t* w, int ldw, int N, int K, const float* bias,
             const half_t* res, int ldr, int epi_extra, void* y, int ldy, Act* stat, int rows_per_sample, const half_t* w_tm) {
  GemmArgs g;
  g.A0 = a.p0; g.lda0 = a.ld0;
  if (a.p1 && K > a.C0) { g.A1 = a.p1; g.lda1 = a.ld1; g.K0 = a.C0; }
  g.Wt = w; g.ldw = ldw; g.M = (int)M; g.N = N; g.K = K;
  if (w_tm) { g.Wt = w_tm; g.ldw = 64; g.w_kts = N * 128; }
  g.epi = epi_extra;
  if (bias) { g.epi |= EPI_BIAS_N; g.bias = bias; }
  if (res) { g.epi |= EPI_RESIDUAL; g.R = res; g.ldr = ldr; }
  g.C = y; g.ldc = ldy;
  g.rows_per_sample_hint = rows_per_sample;
  gn_emit(ctx, g, stat, rows_per_sample);
  return launch_gemm(ctx, g);
}

// q/k/v projection as one GEMM with a transposed tail (GemmArgs::Vt): whole 32-token passes per sample, V columns on a tile boundary
// The graph-shape switches live in the context (TsdOptions); a setter changes THAT context only and returns the old value.
extern "C" int tsd_debug_set_qkv_fuse(tsd_ctx* ctx, int on) {
  return ctx_set_option(ctx, &TsdOptions::qkv_fuse, on, 0, 1);
}
static bool qkv_fused_ok(const tsd_ctx* ctx, int S, int Sp, int C) {
  const int BN = ((3 * C) % 160 == 0) ? 160 : 128;
  return ctx->opt.qkv_fuse && S % 32 == 0 && Sp == S && (2 * C) % BN == 0;
}

// context K / V^T of all attention blocks from one GEMM: the concatenated v_proj rows follow the k_proj rows in the blob
static bool ctx_kv_fused_ok(const tsd_ctx* ctx, const UNetW& u, int Tp) {
  const int CK = u.kproj_all.N;
  return ctx->opt.qkv_fuse && Tp % 8 == 0 && CK % 160 == 0 && u.vproj_all.N == CK && u.vproj_all.Kpad == u.kproj_all.Kpad &&
         u.vproj_all.w == u.kproj_all.w + (int64_t)CK * u.kproj_all.Kpad;
}

static NormSrc norm_src(const CatSrc& x, int C) {
  NormSrc s;
  s.x0 = x.p0; s.ld0 = x.ld0; s.C0 = x.C0;
  if (x.p1 && C > x.C0) { s.x1 = x.p1; s.ld1 = x.ld1; }
  return s;
}

extern "C" int tsd_debug_set_res_fuse_skip(tsd_ctx* ctx, int on) {
  return ctx_set_option(ctx, &TsdOptions::res_fuse_skip, on, 0, 1);
}

// `Unet_Residual_Block.forward` diffusion.mojo:54-72 / VAE `Res_Block.forward` vae.mojo:57-67
int g_resblock(tsd_ctx* ctx, const CatSrc& x, int B, int Hin, int Win, int ups, const ResW& w, const float* tvec,
               int tld, Act& out) {
  const int cin = w.cin, cout = w.cout;
  if (cin % 64 || cout % 64) TSD_FAIL(TSD_E_SHAPE, "residual block: channels (%d,%d) must be multiples of 64", cin, cout);
  if (cin > x.C0 + (x.p1 ? x.C1 : 0)) TSD_FAIL(TSD_E_SHAPE, "residual block: input has fewer than %d channels", cin);
  if (x.p1 && cin > x.C0 && (x.C0 % 64)) TSD_FAIL(TSD_E_SHAPE, "residual block: concat split %d not a multiple of 64", x.C0);
  if (!w.has_skip && (ups || (x.p1 && cin > x.C0))) TSD_FAIL(TSD_E_SHAPE, "residual block: identity skip needs a plain input");
  const int H = ups ? 2 * Hin : Hin, W = ups ? 2 * Win : Win;
  const size_t mark = ctx->arena.mark();
  // GN -> SiLU (only the first cin channels are normalised/consumed: App.A D11)
  Act h = act_alloc(ctx, B, Hin, Win, cin); CHECK_ALLOC(h.p);
  const bool x_stats = !(x.p1 && cin > x.C0) && x.gn_part0 && x.gn_groups0 == w.groups && x.C0 == cin;
  // channel concat (diffusion.mojo:253-270): the norm's groups are sums of whole groups of the two producers' statistics when both
  // were emitted at one granularity that divides the concat's group size - no statistics pass over the concatenated tensor
  GnComposite gc;
  if (ctx->opt.gn_composite && x.p1 && cin == x.C0 + x.C1 && x.gn_part0 && x.gn_part1 && x.gn_groups0 > 0 && x.gn_groups1 > 0 &&
      x.gn_nslab0 == x.gn_nslab1 && x.gn_nslab0 > 0 && x.C0 % x.gn_groups0 == 0 && x.C1 % x.gn_groups1 == 0 && cin % w.groups == 0) {
    const int cf = x.C0 / x.gn_groups0, cpg = cin / w.groups;
    if (cf == x.C1 / x.gn_groups1 && cpg % cf == 0) {
      gc.part0 = x.gn_part0; gc.G0 = x.gn_groups0; gc.part1 = x.gn_part1; gc.G1 = x.gn_groups1; gc.nslab = x.gn_nslab0; gc.comb = cpg / cf;
    }
  }
  TSD_TRY(launch_groupnorm(ctx, norm_src(x, cin), B, Hin * Win, cin, w.groups, w.eps, 1.f, 1, h.p, h.ld,
                           x_stats ? x.gn_part0 : nullptr, x.gn_nslab0, w.gn1.w ? &w.gn1 : nullptr, gc.part0 ? &gc : nullptr));
  Act t1 = act_alloc_gn(ctx, B, H, W, cout, w.groups); CHECK_ALLOC(t1.p);
  TSD_TRY(g_conv3x3(ctx, h, w.conv1, 1, 1, 1, ups, tvec ? tvec + w.time_off : nullptr, tld, nullptr, 0, false, t1.p, t1.ld, &t1));
  Act h3 = act_alloc(ctx, B, H, W, cout); CHECK_ALLOC(h3.p);
  TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(t1), cout), B, H * W, cout, w.groups, w.eps, 1.f, 1, h3.p, h3.ld, t1.gn_part,
                           t1.gn_nslab, w.gn2.w ? &w.gn2 : nullptr));
  Act r;
  // Skip path `x + ...` / `conv1x1(x) + ...` (diffusion.mojo:70-72, vae.mojo:65-67).  At the block's own resolution the 1x1 convolution
  // is folded into the second 3x3 convolution as extra K (one launch and one residual read less, and the products run at the
  // big convolution's rate); behind an upsample it stays a GEMM at the input resolution (a quarter of the rows).
  if (w.has_skip && !ups && ctx->opt.res_fuse_skip > 0 && w.skip.k == 1 && w.skip.Ipad % 64 == 0 && w.skip.Ipad == cin &&
      !(x.p1 && cin > x.C0 && (x.C0 % 64))) {
    TSD_TRY(g_conv3x3(ctx, h3, w.conv2, 1, 1, 1, 0, nullptr, 0, nullptr, 0, false, out.p, out.ld, &out, &x, &w.skip));
    ctx->arena.release(mark);
    return TSD_OK;
  }
  if (w.has_skip) {  // 1x1 conv on the raw input, at the INPUT resolution (commutes with nearest upsample)
    r = act_alloc(ctx, B, Hin, Win, cout); CHECK_ALLOC(r.p);
    TSD_TRY(g_linear(ctx, x, (int64_t)B * Hin * Win, w.skip.w, w.skip.Ipad, cout, w.skip.Ipad, w.skip.b, nullptr, 0, 0,
                     r.p, r.ld));
  } else {
    r.p = const_cast<half_t*>(x.p0); r.ld = x.ld0; r.B = B; r.H = Hin; r.W = Win; r.C = cout;
  }
  TSD_TRY(g_conv3x3(ctx, h3, w.conv2, 1, 1, 1, 0, nullptr, 0, &r, ups, false, out.p, out.ld, &out));
  ctx->arena.release(mark);
  return TSD_OK;
}

// `Unet_Attention_Block.forward` diffusion.mojo:112-147 on NHWC tokens
int g_unet_attn(tsd_ctx* ctx, const Act& x, const AttnW& w, const half_t* ctx16, int T, int Tp, Act& out,
                const CtxKV* pre) {
  const int C = w.C, B = x.B, S = x.H * x.W, d = w.n_embed, Hh = w.n_head;
  const int64_t M = (int64_t)B * S;
  if (C % 64 || x.C != C) TSD_FAIL(TSD_E_SHAPE, "attention block: C=%d (input %d) unsupported", C, x.C);
  if (!attn_fused_supported(d)) TSD_FAIL(TSD_E_SHAPE, "attention block: head dim %d unsupported (40/80/160)", d);
  if (S % 4) TSD_FAIL(TSD_E_SHAPE, "attention block: H*W=%d must be a multiple of 4", S);
  const size_t mark = ctx->arena.mark();
  const float scale = 1.f / sqrtf((float)d);  // helpers/attention.mojo:57-58
  const bool x_stats = x.gn_part && x.gn_groups == 32;
  half_t* tok = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(tok);
  half_t* ln = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(ln);
  half_t* qk = arena_alloc<half_t>(ctx, M * 2 * C); CHECK_ALLOC(qk);
  const int Sp = round_up(S, 8);
  half_t* vt = arena_alloc<half_t>(ctx, (int64_t)B * C * Sp); CHECK_ALLOC(vt);
  half_t* ao = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(ao);
  CatSrc a;
  // The block's head - GroupNorm-apply, conv_in, LayerNorm, q / k / V^T projections - is local to a token row as well once
  // the GroupNorm statistics are known (the producer's epilogue emitted them): one kernel at the 64x64 level.
  const half_t* head_stream = w.head_stream;
  const bool head_ok = Sp == S && x.ld % 8 == 0 && attn_tail_supported(ctx, C, d, Hh, 1, M, S) && attn_head_weights_ok(w);
  if (head_ok && !head_stream && !pre) {  // block-level entry: no model, pack on the fly
    half_t* hs = arena_alloc<half_t>(ctx, (int64_t)attn_head_stream_bytes() / 2); CHECK_ALLOC(hs);
    TSD_TRY(launch_attn_head_pack(ctx, w.conv_in.w, w.conv_in.Ipad, w.sa_in.w, w.sa_in.Kpad, hs));
    head_stream = hs;
  }
  if (head_ok && head_stream) {
    float* st = arena_alloc<float>(ctx, (int64_t)B * 32 * 2); CHECK_ALLOC(st);
    if (x_stats) TSD_TRY(launch_gn_finalize(ctx, x.gn_part, x.gn_nslab, B, S, C, 32, 1e-6f, 1.f, st));  // :89
    else TSD_TRY(launch_gn_stats(ctx, x.p, x.ld, B, S, C, 32, 1e-6f, 1.f, st));  // no producer statistics: one read of x
    AttnHeadArgs ha;
    ha.x = x.p; ha.ld_x = x.ld; ha.gn_stats = st; ha.wstream = head_stream; ha.b_in = w.conv_in.b;
    ha.tok = tok; ha.ld_tok = C; ha.qk = qk; ha.ld_qk = 2 * C; ha.vt = vt; ha.ld_vt = Sp; ha.s_vt = (int64_t)C * Sp;
    ha.M = M; ha.S = S; ha.eps = 1e-5f;
    TSD_TRY(launch_attn_head(ctx, ha));
  } else {
    half_t* h0 = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(h0);
    TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(x), C), B, S, C, 32, 1e-6f, 1.f, 0, h0, C, x_stats ? x.gn_part : nullptr,
                             x.gn_nslab, w.gn.w ? &w.gn : nullptr));  // :89,:116
    a.p0 = h0; a.ld0 = C; a.C0 = C;
    TSD_TRY(g_linear(ctx, a, M, w.conv_in.w, w.conv_in.Ipad, C, C, w.conv_in.b, nullptr, 0, 0, tok, C, nullptr, S, w.conv_in.w_tm));  // :117
    if (Sp != S) TSD_TRY(zero_async(ctx, vt, (size_t)B * C * Sp * sizeof(half_t)));  // pad keys must be finite (P = 0 there)
    // ---- self attention (:122-126) ----
    TSD_TRY(launch_layernorm(ctx, tok, M, C, C, 1e-5f, ln, C, w.ln[0].w ? &w.ln[0] : nullptr));
    a.p0 = ln; a.ld0 = C; a.C0 = C;
    if (qkv_fused_ok(ctx, S, Sp, C)) {  // q | k token-major and V^T channel-major from ONE GEMM over in_proj's 3C rows (transposed tail)
      GemmArgs g;
      g.A0 = ln; g.lda0 = C; g.Wt = w.sa_in.w; g.ldw = w.sa_in.Kpad; g.M = (int)M; g.N = 3 * C; g.K = C;
      if (w.sa_in.w_tm) { g.Wt = w.sa_in.w_tm; g.ldw = 64; g.w_kts = 3 * C * 128; }
      g.C = qk; g.ldc = 2 * C; g.rows_per_sample_hint = S;
      g.Vt = vt; g.vt_n0 = 2 * C; g.vt_ld = Sp; g.vt_S = S; g.vt_sB = (int64_t)C * Sp;
      TSD_TRY(launch_gemm(ctx, g));
    } else {
    TSD_TRY(g_linear(ctx, a, M, w.sa_in.w, w.sa_in.Kpad, 2 * C, C, nullptr, nullptr, 0, 0, qk, 2 * C, nullptr, S));  // q,k
    {  // V^T[b] = W_v . ln_b^T  -> [B][C][S]
      GemmArgs g;
      g.A0 = w.sa_in.w + (int64_t)2 * C * w.sa_in.Kpad; g.lda0 = w.sa_in.Kpad; g.sA = 0;
      g.Wt = ln; g.ldw = C; g.sW = (int64_t)S * C;
      g.M = C; g.N = S; g.K = C; g.batch = B;
      g.C = vt; g.ldc = Sp; g.sC = (int64_t)C * Sp;
      TSD_TRY(launch_gemm(ctx, g));
    }
    }
  }
  a.ld0 = C; a.C0 = C;
  AttnArgs fa;
  fa.Q = qk; fa.ldq = 2 * C; fa.sQ = (int64_t)S * 2 * C;
  fa.K = qk + C; fa.ldk = 2 * C; fa.sK = (int64_t)S * 2 * C;
  fa.Vt = vt; fa.ldvt = Sp; fa.sVt = (int64_t)C * Sp;
  fa.O = ao; fa.ldo = C; fa.sO = (int64_t)S * C;
  fa.B = B; fa.H = Hh; fa.d = d; fa.Sq = S; fa.Sk = S; fa.scale = scale;
  TSD_TRY(launch_flash_attention(ctx, fa));
  // context keys / values of this block: projected once for all blocks by g_unet_forward, or here (block-level entry)
  CtxKV kv;
  if (pre) kv = *pre;  // context K / V^T of all nine blocks were projected in two batched GEMMs (g_unet_forward)
  else {
    half_t* kc = arena_alloc<half_t>(ctx, (int64_t)B * Tp * C); CHECK_ALLOC(kc);
    half_t* vtc = arena_alloc<half_t>(ctx, (int64_t)B * C * Tp); CHECK_ALLOC(vtc);
    kv.K = kc; kv.ldk = C; kv.sK = (int64_t)Tp * C; kv.Vt = vtc; kv.ldvt = Tp; kv.sVt = (int64_t)C * Tp;
    GemmArgs g;  // K_c[b] = ctx_b . W_k^T -> [B][Tp][C]
    g.A0 = ctx16; g.lda0 = w.ca_k.Kpad; g.sA = (int64_t)Tp * w.ca_k.Kpad;
    g.Wt = w.ca_k.w; g.ldw = w.ca_k.Kpad;
    g.M = Tp; g.N = C; g.K = w.ca_k.Kpad; g.batch = B;
    g.C = kc; g.ldc = C; g.sC = (int64_t)Tp * C;
    TSD_TRY(launch_gemm(ctx, g));
    GemmArgs v;  // V_c^T[b] = W_v . ctx_b^T -> [B][C][Tp]
    v.A0 = w.ca_v.w; v.lda0 = w.ca_v.Kpad; v.sA = 0;
    v.Wt = ctx16; v.ldw = w.ca_v.Kpad; v.sW = (int64_t)Tp * w.ca_v.Kpad;
    v.M = C; v.N = Tp; v.K = w.ca_v.Kpad; v.batch = B;
    v.C = vtc; v.ldc = Tp; v.sC = (int64_t)C * Tp;
    TSD_TRY(launch_gemm(ctx, v));
  }
  // Everything after the self-attention core is local to a token row: at the 64x64 level (C = 320) it runs as ONE kernel
  // (kernels_chain.hip) instead of nine launches.  Reference norms / tanh-GELU only (the torch-norm extension keeps the
  // op-by-op path).
  // The block-level entry (no model) packs the stream into the workspace on the fly.
  const half_t* tail_stream = w.tail_stream;
  const bool tail_ok = attn_tail_supported(ctx, C, d, Hh, T, M, S) && attn_tail_weights_ok(w);
  if (tail_ok && !tail_stream && !pre) {
    half_t* ts = arena_alloc<half_t>(ctx, (int64_t)attn_tail_stream_bytes() / 2); CHECK_ALLOC(ts);
    TSD_TRY(launch_attn_tail_pack(ctx, w.sa_out.w, w.sa_out.Kpad, w.ca_q.w, w.ca_q.Kpad, w.ca_out.w, w.ca_out.Kpad, w.geglu1.w,
                                  w.geglu1.Kpad, w.geglu2.w, w.geglu2.Kpad, w.conv_out.w, w.conv_out.Ipad, ts));
    tail_stream = ts;
  }
  if (tail_ok && tail_stream) {
    AttnTailArgs ta;
    ta.ao = ao; ta.ld_ao = C; ta.tok = tok; ta.ld_tok = C; ta.x = x.p; ta.ld_x = x.ld; ta.out = out.p; ta.ld_out = out.ld;
    ta.wstream = tail_stream;
    ta.bso = w.sa_out.b; ta.bco = w.ca_out.b; ta.b1 = w.geglu1.b; ta.b2 = w.geglu2.b; ta.bout = w.conv_out.b;
    ta.Kc = kv.K; ta.ldk = kv.ldk; ta.sK = kv.sK; ta.Vt = kv.Vt; ta.ldvt = kv.ldvt; ta.sVt = kv.sVt;
    ta.C = C; ta.d = d; ta.heads = Hh; ta.T = T; ta.S = S; ta.M = M; ta.scale = scale; ta.eps = 1e-5f;
    if (out.gn_buf && out.gn_groups == 32) {  // statistics for the next block's GroupNorm(32): one slab per 32 rows
      ta.gn_part = out.gn_buf; ta.gn_nslab = S / 32;
      out.gn_part = out.gn_buf; out.gn_nslab = S / 32;
    }
    TSD_TRY(launch_attn_tail(ctx, ta));
    ctx->arena.release(mark);
    return TSD_OK;
  }
  half_t* tok2 = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(tok2);
  a.p0 = ao;
  TSD_TRY(g_linear(ctx, a, M, w.sa_out.w, w.sa_out.Kpad, C, C, w.sa_out.b, tok, C, 0, tok2, C, nullptr, S, w.sa_out.w_tm));
  // ---- cross attention (:129-133) ----
  TSD_TRY(launch_layernorm(ctx, tok2, M, C, C, 1e-5f, ln, C, w.ln[1].w ? &w.ln[1] : nullptr));
  half_t* q = qk;  // reuse
  a.p0 = ln;
  TSD_TRY(g_linear(ctx, a, M, w.ca_q.w, w.ca_q.Kpad, C, C, nullptr, nullptr, 0, 0, q, C, nullptr, S, w.ca_q.w_tm));
  fa.Q = q; fa.ldq = C; fa.sQ = (int64_t)S * C;
  fa.K = kv.K; fa.ldk = kv.ldk; fa.sK = kv.sK;
  fa.Vt = kv.Vt; fa.ldvt = kv.ldvt; fa.sVt = kv.sVt;
  fa.Sk = T;
  TSD_TRY(launch_flash_attention(ctx, fa));
  half_t* tok3 = tok;  // tok (first residual) is dead after tok2 was produced
  a.p0 = ao;
  TSD_TRY(g_linear(ctx, a, M, w.ca_out.w, w.ca_out.Kpad, C, C, w.ca_out.b, tok2, C, 0, tok3, C, nullptr, S, w.ca_out.w_tm));
  // ---- GEGLU feed-forward (:136-143) ----
  TSD_TRY(launch_layernorm(ctx, tok3, M, C, C, 1e-5f, ln, C, w.ln[2].w ? &w.ln[2] : nullptr));
  half_t* gg = arena_alloc<half_t>(ctx, M * 4 * C); CHECK_ALLOC(gg);
  a.p0 = ln;
  if (w.gelu_erf) {  // real-checkpoint extension: torch's exact GELU, as a pass of its own (the default path stays fused)
    half_t* pre = arena_alloc<half_t>(ctx, M * 8 * C); CHECK_ALLOC(pre);
    TSD_TRY(g_linear(ctx, a, M, w.geglu1.w, w.geglu1.Kpad, 8 * C, C, w.geglu1.b, nullptr, 0, 0, pre, 8 * C, nullptr, S));
    TSD_TRY(launch_geglu_erf_f16(ctx, pre, M, 4 * C, gg));
  } else {
    TSD_TRY(g_linear(ctx, a, M, w.geglu1.w, w.geglu1.Kpad, 8 * C, C, w.geglu1.b, nullptr, 0, EPI_GEGLU, gg, 4 * C, nullptr, S, w.geglu1.w_tm));
  }
  CatSrc ag; ag.p0 = gg; ag.ld0 = 4 * C; ag.C0 = 4 * C;
  if (w.fold_w && ctx->opt.fold_out) {
    // GEGLU's second linear + residual (:143) and the output 1x1 conv + long residual (:146) are linear in [h | tok3]: ONE GEMM over the
    // channel concat with the weights folded at model_check_ready (AttnW::fold_w) - a launch and an M x C round trip fewer per block
    ag.p1 = tok3; ag.ld1 = C; ag.C1 = C;
    TSD_TRY(g_linear(ctx, ag, M, w.fold_w, 5 * C, C, 5 * C, w.fold_b, x.p, x.ld, 0, out.p, out.ld, &out, S, w.fold_w_tm));
    ctx->arena.release(mark);
    return TSD_OK;
  }
  half_t* tok4 = tok2;  // tok2 is dead after tok3
  TSD_TRY(g_linear(ctx, ag, M, w.geglu2.w, w.geglu2.Kpad, C, 4 * C, w.geglu2.b, tok3, C, 0, tok4, C, nullptr, S, w.geglu2.w_tm));
  // ---- output 1x1 conv + long residual (:146) ----
  a.p0 = tok4;
  TSD_TRY(g_linear(ctx, a, M, w.conv_out.w, w.conv_out.Ipad, C, C, w.conv_out.b, x.p, x.ld, 0, out.p, out.ld, &out, S, w.conv_out.w_tm));
  ctx->arena.release(mark);
  return TSD_OK;
}

// Attention core on projected q/k (token-major) and v^T (channel-major): fused flash kernel for the UNet
// head dims, otherwise (one head, d % 64 == 0: the VAE's 512-wide head) scores are materialised like the
// reference does (helpers/attention.mojo:46) - batched GEMM -> row softmax -> batched GEMM.
int g_attn_core(tsd_ctx* ctx, const AttnArgs& fa) {
  if (attn_fused_supported(fa.d)) return launch_flash_attention(ctx, fa);
  if (fa.H != 1 || fa.d % 64 || fa.Sk % 64)
    TSD_FAIL(TSD_E_SHAPE, "attention: heads=%d d_head=%d Tk=%d unsupported (fused: d in {40,80,160}; unfused: 1 head, "
             "d%%64==0, Tk%%64==0)", fa.H, fa.d, fa.Sk);
  const int B = fa.B, Sq = fa.Sq, Sk = fa.Sk, C = fa.d;
  const size_t mark = ctx->arena.mark();
  half_t* sc = arena_alloc<half_t>(ctx, (int64_t)B * Sq * Sk); CHECK_ALLOC(sc);
  {
    GemmArgs g;  // scores[b] = q_b k_b^T * scale
    g.A0 = fa.Q; g.lda0 = fa.ldq; g.sA = fa.sQ;
    g.Wt = fa.K; g.ldw = fa.ldk; g.sW = fa.sK;
    g.M = Sq; g.N = Sk; g.K = C; g.batch = B;
    g.out_scale = fa.scale;
    g.C = sc; g.ldc = Sk; g.sC = (int64_t)Sq * Sk;
    TSD_TRY(launch_gemm(ctx, g));
  }
  TSD_TRY(launch_softmax_rows_f16(ctx, sc, (int64_t)B * Sq, Sk, Sk));
  {
    GemmArgs g;  // o_b = P_b v_b
    g.A0 = sc; g.lda0 = Sk; g.sA = (int64_t)Sq * Sk;
    g.Wt = fa.Vt; g.ldw = fa.ldvt; g.sW = fa.sVt;
    g.M = Sq; g.N = C; g.K = Sk; g.batch = B;
    g.C = fa.O; g.ldc = fa.ldo; g.sC = fa.sO;
    TSD_TRY(launch_gemm(ctx, g));
  }
  ctx->arena.release(mark);
  return TSD_OK;
}

// q,k (token-major [B*S][2C]) and v^T ([B][C][Sp]) from a fused in_proj (3C, C) (helpers/attention.mojo:29)
int g_qkv_proj(tsd_ctx* ctx, const half_t* x, int B, int S, int C, const LinW& in_proj, half_t* qk, half_t* vt,
               int Sp) {
  if (qkv_fused_ok(ctx, S, Sp, C) && in_proj.Kpad == C) {
    GemmArgs g;
    g.A0 = x; g.lda0 = C; g.Wt = in_proj.w; g.ldw = in_proj.Kpad; g.M = B * S; g.N = 3 * C; g.K = in_proj.Kpad;
    if (in_proj.b) { g.epi = EPI_BIAS_N; g.bias = in_proj.b; }
    g.C = qk; g.ldc = 2 * C; g.rows_per_sample_hint = S;
    g.Vt = vt; g.vt_n0 = 2 * C; g.vt_ld = Sp; g.vt_S = S; g.vt_sB = (int64_t)C * Sp;
    return launch_gemm(ctx, g);
  }
  CatSrc a; a.p0 = x; a.ld0 = C; a.C0 = C;
  TSD_TRY(g_linear(ctx, a, (int64_t)B * S, in_proj.w, in_proj.Kpad, 2 * C, in_proj.Kpad, in_proj.b, nullptr, 0, 0, qk,
                   2 * C));
  GemmArgs g;  // V^T[b] = W_v . x_b^T
  g.A0 = in_proj.w + (int64_t)2 * C * in_proj.Kpad; g.lda0 = in_proj.Kpad; g.sA = 0;
  g.Wt = x; g.ldw = C; g.sW = (int64_t)S * C;
  g.M = C; g.N = S; g.K = in_proj.Kpad; g.batch = B;
  if (in_proj.b) { g.epi = EPI_BIAS_M; g.bias = in_proj.b + 2 * C; }
  g.C = vt; g.ldc = Sp; g.sC = (int64_t)C * Sp;
  return launch_gemm(ctx, g);
}

// VAE `Attention_Block.forward` vae.mojo:17-27: GroupNorm(32) -> one head of width C -> + x
int g_vae_attn(tsd_ctx* ctx, const Act& x, const VaeAttnW& w, Act& out) {
  const int C = w.C, B = x.B, S = x.H * x.W;
  const int64_t M = (int64_t)B * S;
  if (C % 64 || S % 8) TSD_FAIL(TSD_E_SHAPE, "vae attention: C=%d, H*W=%d unsupported", C, S);
  const size_t mark = ctx->arena.mark();
  half_t* h0 = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(h0);
  const bool x_stats = x.gn_part && x.gn_groups == 32;
  TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(x), C), B, S, C, 32, w.eps, 1.f, 0, h0, C, x_stats ? x.gn_part : nullptr, x.gn_nslab,
                           w.gn.w ? &w.gn : nullptr));
  half_t* qk = arena_alloc<half_t>(ctx, M * 2 * C); CHECK_ALLOC(qk);
  half_t* vt = arena_alloc<half_t>(ctx, (int64_t)B * C * S); CHECK_ALLOC(vt);
  half_t* ao = arena_alloc<half_t>(ctx, M * C); CHECK_ALLOC(ao);
  TSD_TRY(g_qkv_proj(ctx, h0, B, S, C, w.in_proj, qk, vt, S));
  AttnArgs fa;
  fa.Q = qk; fa.ldq = 2 * C; fa.sQ = (int64_t)S * 2 * C;
  fa.K = qk + C; fa.ldk = 2 * C; fa.sK = (int64_t)S * 2 * C;
  fa.Vt = vt; fa.ldvt = S; fa.sVt = (int64_t)C * S;
  fa.O = ao; fa.ldo = C; fa.sO = (int64_t)S * C;
  fa.B = B; fa.H = 1; fa.d = C; fa.Sq = S; fa.Sk = S; fa.scale = 1.f / sqrtf((float)C);
  TSD_TRY(g_attn_core(ctx, fa));
  CatSrc a; a.p0 = ao; a.ld0 = C; a.C0 = C;
  TSD_TRY(g_linear(ctx, a, M, w.out_proj.w, w.out_proj.Kpad, C, C, w.out_proj.b, x.p, x.ld, 0, out.p, out.ld, &out, S));
  ctx->arena.release(mark);
  return TSD_OK;
}

// ---- `Diffusion.forward` diffusion.mojo:309-318 -------------------------------------------------------
static int g_unet_full_forward(tsd_model* m, const float* latents_chw, const half_t* ctx16, int T, int Tp,
                               const float* temb, int B, int L, float* eps_out_chw, bool eps_nhwc);

int g_unet_forward(tsd_model* m, const float* latents_chw, const half_t* ctx16, int T, int Tp, const float* temb, int B,
                   int L, float* eps_out_chw, bool eps_nhwc) {
  if (is_full_unet_kind(m->kind)) {
    const int prev = gemm_set_splitk_big(m->ctx, 4);  // measured for this graph at its batch of 4 (BASELINE configs[4]): +4.3 %
    const int r = g_unet_full_forward(m, latents_chw, ctx16, T, Tp, temb, B, L, eps_out_chw, eps_nhwc);
    gemm_set_splitk_big(m->ctx, prev);
    return r;
  }
  tsd_ctx* ctx = m->ctx;
  const UNetW& u = m->unet;
  if (L % 8) TSD_FAIL(TSD_E_SHAPE, "UNet: latent side %d must be a multiple of 8", L);
  // ---- time path (diffusion.mojo:17-21 and :61-62 for all nine residual blocks) ----
  float* t1 = arena_alloc<float>(ctx, (int64_t)B * 1280); CHECK_ALLOC(t1);
  float* time = arena_alloc<float>(ctx, (int64_t)B * 1280); CHECK_ALLOC(time);
  float* tvec = arena_alloc<float>(ctx, (int64_t)B * u.tproj.N); CHECK_ALLOC(tvec);
  TSD_TRY(launch_small_linear(ctx, temb, B, 320, 320, u.t1.w, u.t1.Kpad, u.t1.b, 1280, 0, t1, 1280));
  TSD_TRY(launch_small_linear(ctx, t1, B, 1280, 1280, u.t2.w, u.t2.Kpad, u.t2.b, 1280, 1, time, 1280));
  TSD_TRY(launch_small_linear(ctx, time, B, 1280, 1280, u.tproj.w, u.tproj.Kpad, u.tproj.b, u.tproj.N, 1, tvec,
                              u.tproj.N));
  const int tld = u.tproj.N;
  // ---- input ----
  // `Conv2D(4, 320, 3)` (diffusion.mojo:236): the latent's 4 channels padded to a 64-channel NHWC tensor made the implicit GEMM walk
  // nine K tiles with 4 of 64 columns live; gathered to im2col rows at the boundary (36 of 64 columns live) it is ONE K tile
  const bool in_im2col = ctx->opt.conv_in_im2col && u.conv_in_im2col && u.conv1.I == 4;
  Act x0 = act_alloc(ctx, B, L, L, 64); CHECK_ALLOC(x0.p);
  if (in_im2col) TSD_TRY(launch_chw_f32_to_im2col3x3_f16(ctx, latents_chw, B, 4, L, L, x0.p));
  else TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, latents_chw, B, 4, L, L, 4, 1.f, x0.p, 64));
  Act a[24];
  // every layer output feeds a GroupNorm (the next block's first norm: 32 groups; the output layer's: 320), so it is
  // allocated with room for the statistics its producer's epilogue emits
  auto alloc_out = [&](int i, int side, int C) -> int {
    // layers 11 and 16 feed GroupNorm(32) over a concat with a narrower skip (1280 + 640, 640 + 320): statistics in the skip's granularity
    const int groups = i == 23 ? 320 : (i == 11 ? concat_stat_groups(C, 640, 32) : (i == 16 ? concat_stat_groups(C, 320, 32) : 32));
    a[i] = act_alloc_gn(ctx, B, side, side, C, groups);
    CHECK_ALLOC(a[i].p);
    return TSD_OK;
  };
  const int L1 = L / 2, L2 = L / 4;
  auto res = [&](int i, const CatSrc& src, int side_in, int ups) -> int {
    const ResW& w = u.res[i - 1];
    TSD_TRY(alloc_out(i, ups ? side_in * 2 : side_in, w.cout));
    return g_resblock(ctx, src, B, side_in, side_in, ups, w, tvec, tld, a[i]);
  };
  // context K and V^T projections of all nine attention blocks in two GEMMs (helpers/attention.mojo:102-103;
  // the k_proj / v_proj weights are contiguous in the blob): K_all [B*Tp][6720], V^T_all [B][6720][Tp]
  const int CK = u.kproj_all.N;
  half_t* kc_all = arena_alloc<half_t>(ctx, (int64_t)B * Tp * CK); CHECK_ALLOC(kc_all);
  half_t* vtc_all = arena_alloc<half_t>(ctx, (int64_t)B * CK * Tp); CHECK_ALLOC(vtc_all);
  if (ctx_kv_fused_ok(ctx, u, Tp)) {  // k_proj | v_proj rows are adjacent in the blob: one GEMM, the V half stored transposed (GemmArgs::Vt)
    GemmArgs g;
    g.A0 = ctx16; g.lda0 = u.kproj_all.Kpad; g.Wt = u.kproj_all.w; g.ldw = u.kproj_all.Kpad;
    g.M = B * Tp; g.N = 2 * CK; g.K = u.kproj_all.Kpad; g.C = kc_all; g.ldc = CK;
    if (u.kproj_all.w_tm) { g.Wt = u.kproj_all.w_tm; g.ldw = 64; g.w_kts = 2 * CK * 128; }
    g.Vt = vtc_all; g.vt_n0 = CK; g.vt_ld = Tp; g.vt_S = Tp; g.vt_sB = (int64_t)CK * Tp;
    TSD_TRY(launch_gemm(ctx, g));
  } else {
    GemmArgs g;
    g.A0 = ctx16; g.lda0 = u.kproj_all.Kpad; g.Wt = u.kproj_all.w; g.ldw = u.kproj_all.Kpad;
    g.M = B * Tp; g.N = CK; g.K = u.kproj_all.Kpad; g.C = kc_all; g.ldc = CK;
    TSD_TRY(launch_gemm(ctx, g));
    GemmArgs v;
    v.A0 = u.vproj_all.w; v.lda0 = u.vproj_all.Kpad; v.sA = 0;
    v.Wt = ctx16; v.ldw = u.vproj_all.Kpad; v.sW = (int64_t)Tp * u.vproj_all.Kpad;
    v.M = CK; v.N = Tp; v.K = u.vproj_all.Kpad; v.batch = B;
    v.C = vtc_all; v.ldc = Tp; v.sC = (int64_t)CK * Tp;
    TSD_TRY(launch_gemm(ctx, v));
  }
  auto attn = [&](int i) -> int {
    TSD_TRY(alloc_out(i, a[i - 1].H, a[i - 1].C));
    const AttnW& w = u.attn[i - 1];
    CtxKV kv;
    kv.K = kc_all + w.kv_off; kv.ldk = CK; kv.sK = (int64_t)Tp * CK;
    kv.Vt = vtc_all + (int64_t)w.kv_off * Tp; kv.ldvt = Tp; kv.sVt = (int64_t)CK * Tp;
    return g_unet_attn(ctx, a[i - 1], w, ctx16, T, Tp, a[i], &kv);
  };
  // encoders (diffusion.mojo:236-250)
  TSD_TRY(alloc_out(1, L, 320));
  if (in_im2col) TSD_TRY(g_linear(ctx, cat1(x0), (int64_t)B * L * L, u.conv_in_im2col, 64, u.conv1.Opad, 64, u.conv1.b, nullptr, 0, 0, a[1].p,
                                  a[1].ld, &a[1], L * L));
  else TSD_TRY(g_conv3x3(ctx, x0, u.conv1, 1, 1, 1, 0, nullptr, 0, nullptr, 0, false, a[1].p, a[1].ld, &a[1]));
  TSD_TRY(res(2, cat1(a[1]), L, 0));
  TSD_TRY(attn(3));
  TSD_TRY(alloc_out(4, L1, 320));
  TSD_TRY(g_conv3x3(ctx, a[3], u.conv4, 2, 1, 1, 0, nullptr, 0, nullptr, 0, false, a[4].p, a[4].ld, &a[4]));
  TSD_TRY(res(5, cat1(a[4]), L1, 0));
  TSD_TRY(attn(6));
  TSD_TRY(alloc_out(7, L2, 640));
  TSD_TRY(g_conv3x3(ctx, a[6], u.conv7, 2, 1, 1, 0, nullptr, 0, nullptr, 0, false, a[7].p, a[7].ld, &a[7]));
  TSD_TRY(res(8, cat1(a[7]), L2, 0));
  TSD_TRY(attn(9));
  // decoders (diffusion.mojo:253-272); skip4 / skip2 are dead (App.A D11): layers 15 and 20 declare
  // fewer input channels than the concat provides and only read the first in_channels.
  TSD_TRY(res(10, cat2(a[9], a[9]), L2, 0));
  TSD_TRY(attn(11));
  TSD_TRY(res(12, cat2(a[11], a[7]), L2, 0));
  TSD_TRY(attn(13));
  TSD_TRY(res(15, cat1(a[13]), L2, 1));  // layer14 Upsample folded into the conv/residual addressing
  a[14] = a[15];
  TSD_TRY(attn(16));
  TSD_TRY(res(17, cat2(a[16], a[4]), L1, 0));
  TSD_TRY(attn(18));
  TSD_TRY(res(20, cat1(a[18]), L1, 1));  // layer19 Upsample folded
  a[19] = a[20];
  TSD_TRY(attn(21));
  TSD_TRY(res(22, cat2(a[21], a[1]), L, 0));
  TSD_TRY(attn(23));
  // `UNet_Output_Layer` diffusion.mojo:287-291: GroupNorm(320 groups) -> SiLU -> Conv3x3(320,4)
  Act hf = act_alloc(ctx, B, L, L, 320); CHECK_ALLOC(hf.p);
  TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(a[23]), 320), B, L * L, 320, 320, 1e-5f, 1.f, 1, hf.p, hf.ld,
                           a[23].gn_groups == 320 ? a[23].gn_part : nullptr, a[23].gn_nslab));
  if (eps_nhwc && u.final_conv.Opad == 4)  // the caller reads [B][L*L][4] directly
    return g_conv3x3(ctx, hf, u.final_conv, 1, 1, 1, 0, nullptr, 0, nullptr, 0, true, eps_out_chw, 4);
  float* eps_tmp = arena_alloc<float>(ctx, (int64_t)B * L * L * 4); CHECK_ALLOC(eps_tmp);
  TSD_TRY(g_conv3x3(ctx, hf, u.final_conv, 1, 1, 1, 0, nullptr, 0, nullptr, 0, true, eps_tmp, 4));
  TSD_TRY(launch_nhwc_f32_to_chw_f32(ctx, eps_tmp, B, 4, L, L, 4, eps_out_chw));
  return TSD_OK;
}

// Full-size UNet (TSD_MODEL_DIFFUSION_SD15, BASELINE configs[4]): the same blocks driven by the SD15_STEPS table.
// Encoders push their output; every decoder residual block reads concat(x, popped skip) through the two-source views;
// Upsample + conv3x3 is one implicit-GEMM launch that reads its input through the nearest-2x addressing.
static int g_unet_full_forward(tsd_model* m, const float* latents_chw, const half_t* ctx16, int T, int Tp,
                               const float* temb, int B, int L, float* eps_out_chw, bool eps_nhwc) {
  tsd_ctx* ctx = m->ctx;
  const UNetW& u = m->unet;
  if (L % 16) TSD_FAIL(TSD_E_SHAPE, "full-size UNet: latent side %d must be a multiple of 16", L);
  float* t1 = arena_alloc<float>(ctx, (int64_t)B * 1280); CHECK_ALLOC(t1);
  float* time = arena_alloc<float>(ctx, (int64_t)B * 1280); CHECK_ALLOC(time);
  float* tvec = arena_alloc<float>(ctx, (int64_t)B * u.tproj.N); CHECK_ALLOC(tvec);
  TSD_TRY(launch_small_linear(ctx, temb, B, 320, 320, u.t1.w, u.t1.Kpad, u.t1.b, 1280, 0, t1, 1280));
  TSD_TRY(launch_small_linear(ctx, t1, B, 1280, 1280, u.t2.w, u.t2.Kpad, u.t2.b, 1280, 1, time, 1280));
  TSD_TRY(launch_small_linear(ctx, time, B, 1280, 1280, u.tproj.w, u.tproj.Kpad, u.tproj.b, u.tproj.N, 1, tvec,
                              u.tproj.N));
  const int tld = u.tproj.N;
  const bool in_im2col = ctx->opt.conv_in_im2col && u.conv_in_im2col && !u.conv.empty() && u.conv[0].I == 4 && SD15_STEPS[0].l.kind == L_CONV &&
                         SD15_STEPS[0].l.d == 1;  // as in g_unet_forward: the 4-channel input convolution as one im2col K tile
  Act x0 = act_alloc(ctx, B, L, L, 64); CHECK_ALLOC(x0.p);
  if (in_im2col) TSD_TRY(launch_chw_f32_to_im2col3x3_f16(ctx, latents_chw, B, 4, L, L, x0.p));
  else TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, latents_chw, B, 4, L, L, 4, 1.f, x0.p, 64));
  // context K and V^T of all sixteen attention blocks in two GEMMs
  const int CK = u.kproj_all.N;
  half_t* kc_all = arena_alloc<half_t>(ctx, (int64_t)B * Tp * CK); CHECK_ALLOC(kc_all);
  half_t* vtc_all = arena_alloc<half_t>(ctx, (int64_t)B * CK * Tp); CHECK_ALLOC(vtc_all);
  if (ctx_kv_fused_ok(ctx, u, Tp)) {  // k_proj | v_proj rows are adjacent in the blob: one GEMM, the V half stored transposed (GemmArgs::Vt)
    GemmArgs g;
    g.A0 = ctx16; g.lda0 = u.kproj_all.Kpad; g.Wt = u.kproj_all.w; g.ldw = u.kproj_all.Kpad;
    g.M = B * Tp; g.N = 2 * CK; g.K = u.kproj_all.Kpad; g.C = kc_all; g.ldc = CK;
    if (u.kproj_all.w_tm) { g.Wt = u.kproj_all.w_tm; g.ldw = 64; g.w_kts = 2 * CK * 128; }
    g.Vt = vtc_all; g.vt_n0 = CK; g.vt_ld = Tp; g.vt_S = Tp; g.vt_sB = (int64_t)CK * Tp;
    TSD_TRY(launch_gemm(ctx, g));
  } else {
    GemmArgs g;
    g.A0 = ctx16; g.lda0 = u.kproj_all.Kpad; g.Wt = u.kproj_all.w; g.ldw = u.kproj_all.Kpad;
    g.M = B * Tp; g.N = CK; g.K = u.kproj_all.Kpad; g.C = kc_all; g.ldc = CK;
    TSD_TRY(launch_gemm(ctx, g));
    GemmArgs v;
    v.A0 = u.vproj_all.w; v.lda0 = u.vproj_all.Kpad; v.sA = 0;
    v.Wt = ctx16; v.ldw = u.vproj_all.Kpad; v.sW = (int64_t)Tp * u.vproj_all.Kpad;
    v.M = CK; v.N = Tp; v.K = u.vproj_all.Kpad; v.batch = B;
    v.C = vtc_all; v.ldc = Tp; v.sC = (int64_t)CK * Tp;
    TSD_TRY(launch_gemm(ctx, v));
  }
  std::vector<Act> skips;
  Act cur = x0;
  for (int i = 0; i < SD15_N; i++) {
    const UNetStep& st = SD15_STEPS[i];
    const LayerDef& l = st.l;
    // the next consumer of every layer output starts with GroupNorm(32) (the output layer's has 320 groups)
    const int next_groups = i == SD15_N - 1 ? u.final_groups : 32;
    // a decoder layer whose output meets a narrower skip in the next residual block's concat emits its statistics in the skip's
    // granularity (concat_stat_groups), so that block's first GroupNorm needs no statistics pass
    auto stat_groups = [&](int C) -> int {
      if (i + 1 < SD15_N && SD15_STEPS[i + 1].l.kind == L_RES && (SD15_STEPS[i + 1].flags & U_POP) && !(st.flags & U_PUSH) && !skips.empty())
        return concat_stat_groups(C, skips.back().C, 32);
      return next_groups;
    };
    Act y;
    if (l.kind == L_CONV) {
      const int side = cur.H / l.d;
      y = act_alloc_gn(ctx, B, side, side, l.b, next_groups); CHECK_ALLOC(y.p);
      if (i == 0 && in_im2col)
        TSD_TRY(g_linear(ctx, cat1(cur), (int64_t)B * L * L, u.conv_in_im2col, 64, u.conv[0].Opad, 64, u.conv[0].b, nullptr, 0, 0, y.p, y.ld,
                         &y, L * L));
      else
      TSD_TRY(g_conv3x3(ctx, cur, u.conv[i], l.d, 1, 1, 0, nullptr, 0, nullptr, 0, false, y.p, y.ld, &y));
    } else if (l.kind == L_UPCONV) {
      y = act_alloc_gn(ctx, B, cur.H * 2, cur.W * 2, l.b, stat_groups(l.b)); CHECK_ALLOC(y.p);
      TSD_TRY(g_conv3x3(ctx, cur, u.conv[i], 1, 1, 1, 1, nullptr, 0, nullptr, 0, false, y.p, y.ld, &y));
    } else if (l.kind == L_RES) {
      CatSrc src = cat1(cur);
      if (st.flags & U_POP) {
        if (skips.empty()) TSD_FAIL(TSD_E_STATE, "full-size UNet: skip stack underflow at layer %d", i + 1);
        const Act sk = skips.back(); skips.pop_back();
        if (sk.H != cur.H || cur.C + sk.C != l.a) TSD_FAIL(TSD_E_STATE, "full-size UNet: skip mismatch at layer %d", i + 1);
        src = cat2(cur, sk);
      }
      y = act_alloc_gn(ctx, B, cur.H, cur.W, l.b, stat_groups(l.b)); CHECK_ALLOC(y.p);
      TSD_TRY(g_resblock(ctx, src, B, cur.H, cur.W, 0, u.res[i], tvec, tld, y));
    } else {  // L_ATTN
      y = act_alloc_gn(ctx, B, cur.H, cur.W, cur.C, stat_groups(cur.C)); CHECK_ALLOC(y.p);
      const AttnW& w = u.attn[i];
      CtxKV kv;
      kv.K = kc_all + w.kv_off; kv.ldk = CK; kv.sK = (int64_t)Tp * CK;
      kv.Vt = vtc_all + (int64_t)w.kv_off * Tp; kv.ldvt = Tp; kv.sVt = (int64_t)CK * Tp;
      TSD_TRY(g_unet_attn(ctx, cur, w, ctx16, T, Tp, y, &kv));
    }
    cur = y;
    if (st.flags & U_PUSH) skips.push_back(cur);
  }
  Act hf = act_alloc(ctx, B, L, L, 320); CHECK_ALLOC(hf.p);
  TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(cur), 320), B, L * L, 320, u.final_groups, 1e-5f, 1.f, 1, hf.p, hf.ld,
                           cur.gn_groups == u.final_groups ? cur.gn_part : nullptr, cur.gn_nslab,
                           u.final_gn.w ? &u.final_gn : nullptr));
  if (eps_nhwc && u.final_conv.Opad == 4)  // the caller reads [B][L*L][4] directly
    return g_conv3x3(ctx, hf, u.final_conv, 1, 1, 1, 0, nullptr, 0, nullptr, 0, true, eps_out_chw, 4);
  float* eps_tmp = arena_alloc<float>(ctx, (int64_t)B * L * L * 4); CHECK_ALLOC(eps_tmp);
  TSD_TRY(g_conv3x3(ctx, hf, u.final_conv, 1, 1, 1, 0, nullptr, 0, nullptr, 0, true, eps_tmp, 4));
  TSD_TRY(launch_nhwc_f32_to_chw_f32(ctx, eps_tmp, B, 4, L, L, 4, eps_out_chw));
  return TSD_OK;
}

// ---- VAE (vae.mojo:131-159, :221-250) -------------------------------------------------------------------
// The first convolution of either VAE half reads 3 / 4 channels padded to 64: as in the UNet (g_unet_forward) the boundary gathers im2col
// rows and the layer runs as one 64-deep K tile
static bool vae_in_im2col(const tsd_model* m, const LayerDef* layers) {
  return m->ctx->opt.conv_in_im2col && m->vae.conv_in_im2col && layers[0].kind == L_CONV && layers[0].c == 3 && !m->vae.conv.empty() && m->vae.conv[0].I <= 7;
}
static int run_vae(tsd_model* m, const LayerDef* layers, int n_layers, Act cur, bool final_f32, float* out_nhwc_f32,
                   int* out_side, int* out_ld) {
  tsd_ctx* ctx = m->ctx;
  const VaeW& v = m->vae;
  const int B = cur.B;
  int pending_up = 0;
  // groups of the GroupNorm that consumes layer i's output (0: none) - its producer emits the statistics
  auto next_groups = [&](int i) -> int {
    for (int j = i + 1; j < n_layers; j++) {
      const LayerDef& n = layers[j];
      if (n.kind == L_UP || n.kind == L_SILU) continue;
      if (n.kind == L_GN) return n.a;
      if (n.kind == L_RES) return v.res[j].groups;
      if (n.kind == L_ATTN) return 32;
      return 0;
    }
    return 0;
  };
  for (int i = 0; i < n_layers; i++) {
    const LayerDef& l = layers[i];
    const bool last = (i == n_layers - 1);
    if (l.kind == L_UP) { pending_up = 1; continue; }
    if (l.kind == L_SILU) continue;  // fused into the preceding GroupNorm
    if (l.kind == L_GN) {
      Act y = act_alloc(ctx, B, cur.H, cur.W, l.b); CHECK_ALLOC(y.p);
      const int silu = (i + 1 < n_layers && layers[i + 1].kind == L_SILU) ? 1 : 0;
      const bool st = cur.gn_part && cur.gn_groups == l.a && cur.C == l.b;
      TSD_TRY(launch_groupnorm(ctx, norm_src(cat1(cur), l.b), B, cur.H * cur.W, l.b, l.a, v.gn_eps, 1.f, silu, y.p, y.ld,
                               st ? cur.gn_part : nullptr, cur.gn_nslab, v.gn[i].w ? &v.gn[i] : nullptr));
      cur = y;
      continue;
    }
    if (l.kind == L_CONV || l.kind == L_CONV_S2) {
      const ConvW& w = v.conv[i];
      const int stride = l.kind == L_CONV_S2 ? 2 : 1;
      const int pad = l.kind == L_CONV_S2 ? 0 : (l.c == 3 ? 1 : 0);
      const int pad_br = l.kind == L_CONV_S2 ? 1 : pad;  // two_stride_pad (0,1),(0,1), vae.mojo:115-116
      const int side_in = pending_up ? cur.H * 2 : cur.H;
      const int side = l.kind == L_CONV_S2 ? side_in / 2 : side_in;
      if (last && final_f32) {
        if (l.c == 3) TSD_TRY(g_conv3x3(ctx, cur, w, stride, pad, pad_br, pending_up, nullptr, 0, nullptr, 0, true, out_nhwc_f32, w.Opad));
        else {
          GemmArgs g;
          g.A0 = cur.p; g.lda0 = cur.ld; g.Wt = w.w; g.ldw = w.Ipad; g.M = B * side * side; g.N = w.Opad; g.K = w.Ipad;
          g.epi = EPI_BIAS_N | EPI_OUT_F32; g.bias = w.b; g.C = out_nhwc_f32; g.ldc = w.Opad;
          TSD_TRY(launch_gemm(ctx, g));
        }
        *out_side = side; *out_ld = w.Opad;
        return TSD_OK;
      }
      Act y = act_alloc_gn(ctx, B, side, side, w.Opad, next_groups(i)); CHECK_ALLOC(y.p);
      if (i == 0 && vae_in_im2col(m, layers))  // cur holds im2col rows [B*H*W][64]
        TSD_TRY(g_linear(ctx, cat1(cur), (int64_t)B * side * side, v.conv_in_im2col, 64, w.Opad, 64, w.b, nullptr, 0, 0, y.p, y.ld, &y,
                         side * side));
      else if (l.c == 3) TSD_TRY(g_conv3x3(ctx, cur, w, stride, pad, pad_br, pending_up, nullptr, 0, nullptr, 0, false, y.p, y.ld, &y));
      else TSD_TRY(g_linear(ctx, cat1(cur), (int64_t)B * side * side, w.w, w.Ipad, w.Opad, w.Ipad, w.b, nullptr, 0, 0, y.p, y.ld, &y,
                            side * side));
      pending_up = 0;
      cur = y;
      continue;
    }
    if (pending_up) TSD_FAIL(TSD_E_STATE, "vae: upsample must be followed by a conv");
    if (l.kind == L_RES) {
      Act y = act_alloc_gn(ctx, B, cur.H, cur.W, l.b, next_groups(i)); CHECK_ALLOC(y.p);
      TSD_TRY(g_resblock(ctx, cat1(cur), B, cur.H, cur.W, 0, v.res[i], nullptr, 0, y));
      cur = y;
    } else if (l.kind == L_ATTN) {
      Act y = act_alloc_gn(ctx, B, cur.H, cur.W, l.a, next_groups(i)); CHECK_ALLOC(y.p);
      TSD_TRY(g_vae_attn(ctx, cur, v.attn[i], y));
      cur = y;
    }
  }
  return TSD_OK;
}

int g_decoder_forward(tsd_model* m, const float* latents_chw, int B, int L, float* images_chw) {
  tsd_ctx* ctx = m->ctx;
  if (L % 8) TSD_FAIL(TSD_E_SHAPE, "decoder: latent side %d must be a multiple of 8", L);
  Act x0 = act_alloc(ctx, B, L, L, 64); CHECK_ALLOC(x0.p);
  if (vae_in_im2col(m, DECODER_LAYERS)) TSD_TRY(launch_chw_f32_to_im2col3x3_f16(ctx, latents_chw, B, 4, L, L, x0.p, 1.f / 0.18215f));
  else TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, latents_chw, B, 4, L, L, 4, 1.f / 0.18215f, x0.p, 64));  // vae.mojo:222
  const int S = 8 * L;
  float* img = arena_alloc<float>(ctx, (int64_t)B * S * S * 4); CHECK_ALLOC(img);
  int side = 0, ld = 0;
  TSD_TRY(run_vae(m, DECODER_LAYERS, 26, x0, true, img, &side, &ld));
  if (ctx->launch() && (side != S || ld != 4)) TSD_FAIL(TSD_E_STATE, "decoder: unexpected output %d/%d", side, ld);
  TSD_TRY(launch_nhwc_f32_to_chw_f32(ctx, img, B, 3, S, S, 4, images_chw));
  return TSD_OK;
}

int g_encoder_forward(tsd_model* m, const float* images_chw, const float* noise_chw, int B, int S, float* latents_chw) {
  tsd_ctx* ctx = m->ctx;
  if (S % 64) TSD_FAIL(TSD_E_SHAPE, "encoder: image side %d must be a multiple of 64", S);
  Act x0 = act_alloc(ctx, B, S, S, 64); CHECK_ALLOC(x0.p);
  if (vae_in_im2col(m, ENCODER_LAYERS)) TSD_TRY(launch_chw_f32_to_im2col3x3_f16(ctx, images_chw, B, 3, S, S, x0.p));
  else TSD_TRY(launch_chw_f32_to_nhwc_f16(ctx, images_chw, B, 3, S, S, 3, 1.f, x0.p, 64));
  const int L = S / 8;
  float* mom = arena_alloc<float>(ctx, (int64_t)B * L * L * 8); CHECK_ALLOC(mom);
  int side = 0, ld = 0;
  TSD_TRY(run_vae(m, ENCODER_LAYERS, 19, x0, true, mom, &side, &ld));
  if (ctx->launch() && (side != L || ld != 8)) TSD_FAIL(TSD_E_STATE, "encoder: unexpected output %d/%d", side, ld);
  TSD_TRY(launch_encoder_sample(ctx, mom, B, L * L, 8, noise_chw, latents_chw));
  return TSD_OK;
}

// `CLIP.forward` clip.mojo:90-109 on device token ids [B][77] -> fp32 [B][77][768].
// 77 tokens x 12 heads of 64: the attention core runs unfused per sample (head-batched score GEMM -> masked row
// softmax -> head-batched P.V GEMM with keys zero-padded to 128); it is executed once per prompt, off the denoise loop.
int g_clip_forward(tsd_model* m, const int* tokens_dev, int B, float* out_f32) {
  tsd_ctx* ctx = m->ctx;
  const ClipW& c = m->clip;
  const int T = 77, D = 768, H = 12, dh = 64, Tp = 128;
  const int64_t M = (int64_t)B * T;
  half_t* x = arena_alloc<half_t>(ctx, M * D); CHECK_ALLOC(x);
  half_t* x2 = arena_alloc<half_t>(ctx, M * D); CHECK_ALLOC(x2);
  // the GEMM wants N % 4 == 0: the 77 keys are addressed as 80, so the token-major buffers carry 3 zero rows of slack
  // behind the last sample (rows 77..79 of a sample are the next sample's first rows: finite, and masked / multiplied
  // by exactly-zero probabilities)
  const int Tn = 80;
  half_t* ln = arena_alloc<half_t>(ctx, (M + 3) * D); CHECK_ALLOC(ln);
  half_t* qk = arena_alloc<half_t>(ctx, (M + 3) * 2 * D); CHECK_ALLOC(qk);
  TSD_TRY(zero_async(ctx, ln + M * D, (size_t)3 * D * sizeof(half_t)));
  TSD_TRY(zero_async(ctx, qk + M * 2 * D, (size_t)3 * 2 * D * sizeof(half_t)));
  half_t* vt = arena_alloc<half_t>(ctx, (int64_t)B * D * Tp); CHECK_ALLOC(vt);
  half_t* sc = arena_alloc<half_t>(ctx, (int64_t)H * T * Tp); CHECK_ALLOC(sc);
  half_t* ao = arena_alloc<half_t>(ctx, M * D); CHECK_ALLOC(ao);
  half_t* ff = arena_alloc<half_t>(ctx, M * 4 * D); CHECK_ALLOC(ff);
  TSD_TRY(zero_async(ctx, vt, (size_t)B * D * Tp * sizeof(half_t)));  // key padding 77..127 stays zero
  TSD_TRY(launch_clip_embed(ctx, tokens_dev, c.tok, 49408, D, c.pos, B, T, x));  // clip.mojo:17-20
  CatSrc a;
  for (int l = 0; l < 12; l++) {
    const ClipLayerW& w = c.layer[l];
    // ---- LN -> causal self-attention -> + residue (clip.mojo:37-43) ----
    TSD_TRY(launch_layernorm(ctx, x, M, D, D, 1e-5f, ln, D, w.ln1.w ? &w.ln1 : nullptr));
    a.p0 = ln; a.ld0 = D; a.C0 = D;
    TSD_TRY(g_linear(ctx, a, M, w.in_proj.w, w.in_proj.Kpad, 2 * D, D, w.in_proj.b, nullptr, 0, 0, qk, 2 * D));  // q, k
    {  // V^T[b] = W_v . ln_b^T + b_v  -> [B][768][128]
      GemmArgs g;
      g.A0 = w.in_proj.w + (int64_t)2 * D * w.in_proj.Kpad; g.lda0 = w.in_proj.Kpad; g.sA = 0;
      g.Wt = ln; g.ldw = D; g.sW = (int64_t)T * D;
      g.M = D; g.N = Tn; g.K = D; g.batch = B;
      g.epi = EPI_BIAS_M; g.bias = w.in_proj.b + 2 * D;
      g.C = vt; g.ldc = Tp; g.sC = (int64_t)D * Tp;
      TSD_TRY(launch_gemm(ctx, g));
    }
    for (int b = 0; b < B; b++) {
      GemmArgs s;  // scores[h] = q_h k_h^T / sqrt(64)   (helpers/attention.mojo:46,57-58)
      s.A0 = qk + (int64_t)b * T * 2 * D; s.lda0 = 2 * D; s.sA = dh;
      s.Wt = qk + (int64_t)b * T * 2 * D + D; s.ldw = 2 * D; s.sW = dh;
      s.M = T; s.N = Tn; s.K = dh; s.batch = H;
      s.out_scale = 0.125f;
      s.C = sc; s.ldc = Tp; s.sC = (int64_t)T * Tp;
      TSD_TRY(launch_gemm(ctx, s));
      TSD_TRY(launch_softmax_rows_f16_causal(ctx, sc, (int64_t)H * T, T, Tp, T, Tp));
      GemmArgs o;  // o_h = P_h v_h  (keys padded to 128 with zero probabilities / zero V^T columns)
      o.A0 = sc; o.lda0 = Tp; o.sA = (int64_t)T * Tp;
      o.Wt = vt + (int64_t)b * D * Tp; o.ldw = Tp; o.sW = (int64_t)dh * Tp;
      o.M = T; o.N = dh; o.K = Tp; o.batch = H;
      o.C = ao + (int64_t)b * T * D; o.ldc = D; o.sC = dh;
      TSD_TRY(launch_gemm(ctx, o));
    }
    a.p0 = ao; a.ld0 = D; a.C0 = D;
    TSD_TRY(g_linear(ctx, a, M, w.out_proj.w, w.out_proj.Kpad, D, D, w.out_proj.b, x, D, 0, x2, D));
    // ---- LN -> Linear -> quick-GELU -> Linear -> + residue (clip.mojo:44-53) ----
    TSD_TRY(launch_layernorm(ctx, x2, M, D, D, 1e-5f, ln, D, w.ln2.w ? &w.ln2 : nullptr));
    a.p0 = ln;
    TSD_TRY(g_linear(ctx, a, M, w.l4.w, w.l4.Kpad, 4 * D, D, w.l4.b, nullptr, 0, 0, ff, 4 * D));
    TSD_TRY(launch_quick_gelu_f16(ctx, ff, M * 4 * D));
    a.p0 = ff; a.ld0 = 4 * D; a.C0 = 4 * D;
    TSD_TRY(g_linear(ctx, a, M, w.l5.w, w.l5.Kpad, D, 4 * D, w.l5.b, x2, D, 0, x, D));
  }
  TSD_TRY(launch_layernorm(ctx, x, M, D, D, 1e-5f, ln, D, c.final_ln.w ? &c.final_ln : nullptr));  // clip.mojo:106-108
  return launch_f16_to_f32_rows(ctx, ln, M, D, D, out_f32);
}
