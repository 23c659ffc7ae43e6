// graph.h - host-side graph of the hot path: block and module forwards enqueue kernels on the
// context stream; activations live in the workspace arena as NHWC fp16.
#pragma once
#include "model.h"

// channel-concat view of up to two NHWC tensors with identical (B,H,W) (diffusion.mojo:253-270)
struct CatSrc {
  const half_t* p0 = nullptr; int ld0 = 0; int C0 = 0;
  const half_t* p1 = nullptr; int ld1 = 0; int C1 = 0;
  const float* gn_part0 = nullptr; int gn_nslab0 = 0, gn_groups0 = 0;  // producer-emitted GroupNorm statistics of p0
  const float* gn_part1 = nullptr; int gn_nslab1 = 0, gn_groups1 = 0;  // ... and of p1
};
static inline CatSrc cat1(const Act& a) {
  CatSrc c; c.p0 = a.p; c.ld0 = a.ld; c.C0 = a.C;
  c.gn_part0 = a.gn_part; c.gn_nslab0 = a.gn_nslab; c.gn_groups0 = a.gn_groups;
  return c;
}
// Groups to emit statistics for on a C0-channel tensor whose consumer is GroupNorm(groups) over concat(it, a C1-channel skip whose
// own statistics come in C1 / groups-channel groups): the skip's granularity when the concat's groups are whole multiples of it
// (1280 + 640 -> 60-channel groups = 3 x 20: emit 64 groups of 20), else `groups`
static inline int concat_stat_groups(int C0, int C1, int groups) {
  if (C1 <= 0 || C1 % groups || (C0 + C1) % groups) return groups;
  const int cf = C1 / groups, cpg = (C0 + C1) / groups;
  return (cpg % cf == 0 && C0 % cf == 0 && C0 / cf <= 256) ? C0 / cf : groups;
}
// activation whose producer may emit the statistics of the GroupNorm(groups) that will consume it
Act act_alloc_gn(tsd_ctx* ctx, int B, int H, int W, int C, int groups);
static inline CatSrc cat2(const Act& a, const Act& b) {
  CatSrc c = cat1(a); c.p1 = b.p; c.ld1 = b.ld; c.C1 = b.C;
  c.gn_part1 = b.gn_part; c.gn_nslab1 = b.gn_nslab; c.gn_groups1 = b.gn_groups;
  return c;
}

// y = conv3x3(x) (+bias) (+rowvec per sample) (+residual); `ups`: x is read through a nearest-2x upsample.
int g_conv3x3(tsd_ctx* ctx, const Act& x, const ConvW& w, int stride, int pad, int pad_br, int ups,
              const float* rowvec, int rowvec_ld, const Act* res, int res_ups, bool out_f32, void* y, int ldy,
              Act* stat = nullptr,  // stat: the output tensor's Act when GroupNorm statistics are wanted
              const CatSrc* skip_x = nullptr, const ConvW* skip_w = nullptr);  // + conv1x1(skip_x) fused as extra K (same resolution)
// y[M][N] = A[M][K] . W^T (+bias) (+residual) ; A may be a concat view
int g_linear(tsd_ctx* ctx, const CatSrc& a, int64_t M, const half_t* w, int ldw, int N, int K, const float* bias,
             const half_t* res, int ldr, int epi_extra, void* y, int ldy, Act* stat = nullptr, int rows_per_sample = 0,
             const half_t* w_tm = nullptr);  // w_tm: K-tile-major copy of w ([K/64][N][64]) - used instead when given

int g_resblock(tsd_ctx* ctx, const CatSrc& x, int B, int Hin, int Win, int ups, const ResW& w, const float* tvec,
               int tld, Act& out);
struct CtxKV {  // projected context keys (token-major) and values (channel-major) of one attention block
  const half_t* K = nullptr; int ldk = 0; int64_t sK = 0;
  const half_t* Vt = nullptr; int ldvt = 0; int64_t sVt = 0;
};
int g_unet_attn(tsd_ctx* ctx, const Act& x, const AttnW& w, const half_t* ctx16, int T, int Tp, Act& out,
                const CtxKV* pre = nullptr);
int g_vae_attn(tsd_ctx* ctx, const Act& x, const VaeAttnW& w, Act& out);
int g_attn_core(tsd_ctx* ctx, const AttnArgs& fa);
int g_qkv_proj(tsd_ctx* ctx, const half_t* x, int B, int S, int C, const LinW& in_proj, half_t* qk, half_t* vt, int Sp);

// module forwards on device buffers
// latents: fp32 CHW [B,4,L,L]; context16: fp16 [B][Tp][768]; temb: fp32 [B][320]; eps_out: fp32 CHW [B,4,L,L]
// eps_nhwc: eps_out receives the output convolution's own layout, fp32 [B][L*L][4], instead of CHW (the denoise session's DDPM update
// reads it directly: one conversion launch per step less)
int g_unet_forward(tsd_model* m, const float* latents_chw, const half_t* ctx16, int T, int Tp, const float* temb, int B,
                   int L, float* eps_out_chw, bool eps_nhwc = false);
int g_decoder_forward(tsd_model* m, const float* latents_chw, int B, int L, float* images_chw);
int g_encoder_forward(tsd_model* m, const float* images_chw, const float* noise_chw, int B, int S, float* latents_chw);

// run `fn` once in planning mode to size the arena, reserve it, then for real
template <class F>
int run_planned(tsd_ctx* ctx, F&& fn) {
  Arena& a = ctx->arena;
  a.planning = true; a.top = 0; a.peak = 0;
  int r = fn();
  a.planning = false;
  const size_t need = a.peak;
  a.top = 0; a.peak = 0;
  if (r != TSD_OK) return r;
  TSD_TRY(ctx_reserve_arena(ctx, need));
  r = fn();
  a.top = 0;
  return r;
}
// tokens: device int32 [B][77]; out: device fp32 [B][77][768]
int g_clip_forward(tsd_model* m, const int* tokens_dev, int B, float* out_f32);
