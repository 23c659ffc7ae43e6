// api_model.cpp - module-level forwards (device-resident weights), the device-resident denoise
// session (pipeline.mojo:57-127 + sampler.mojo:15-124) and the RCCL weight broadcast.
#include <dlfcn.h>
#include <math.h>
#include <string.h>

#include <vector>

#include "graph.h"

#define NOTNULL(p) \
  if (!(p)) TSD_FAIL(TSD_E_ARG, "%s: argument '%s' is NULL", __func__, #p)

namespace {
int h2d(tsd_ctx* c, void* dst, const void* src, size_t bytes) {
  if (c->launch()) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
  return TSD_OK;
}
int d2h(tsd_ctx* c, void* dst, const void* src, size_t bytes) {
  if (c->launch()) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  return TSD_OK;
}
template <class F>
int run_model(tsd_model* m, F&& fn) {
  tsd_ctx* ctx = m->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  TSD_TRY(model_check_ready(m));
  int r = run_planned(ctx, fn);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  if (r != TSD_OK) return r;
  if (e != hipSuccess) TSD_FAIL(TSD_E_HIP, "stream synchronize failed: %s", hipGetErrorString(e));
  return ctx_check_status(ctx);
}
}  // namespace

extern "C" int tsd_diffusion_forward(tsd_model* m, const float* latents, const float* context, const float* time_emb,
                                     int B, int L, int T, float* out) {
  NOTNULL(m); NOTNULL(latents); NOTNULL(context); NOTNULL(time_emb); NOTNULL(out);
  if (!is_diffusion_kind(m->kind)) TSD_FAIL(TSD_E_ARG, "tsd_diffusion_forward: model is not a Diffusion");
  if (B <= 0 || B > 16 || L <= 0 || T <= 0) TSD_FAIL(TSD_E_SHAPE, "diffusion: B=%d (1..16) L=%d T=%d", B, L, T);
  tsd_ctx* ctx = m->ctx;
  return run_model(m, [&]() -> int {
    const int Tp = round_up(T, 8);
    const int64_t nl = (int64_t)B * 4 * L * L;
    float* dl = arena_alloc<float>(ctx, nl);
    float* dc = arena_alloc<float>(ctx, (int64_t)B * T * 768);
    float* dt = arena_alloc<float>(ctx, (int64_t)B * 320);
    half_t* c16 = arena_alloc<half_t>(ctx, (int64_t)B * Tp * 768);
    float* de = arena_alloc<float>(ctx, nl);
    if (!dl || !dc || !dt || !c16 || !de) TSD_FAIL(TSD_E_ALLOC, "workspace arena exhausted");
    TSD_TRY(h2d(ctx, dl, latents, nl * 4));
    TSD_TRY(h2d(ctx, dc, context, (size_t)B * T * 768 * 4));
    TSD_TRY(h2d(ctx, dt, time_emb, (size_t)B * 320 * 4));
    for (int b = 0; b < B; b++)  // [T][768] -> zero-padded [Tp][768] per sample
      TSD_TRY(launch_f32_to_f16_rows(ctx, dc + (int64_t)b * T * 768, T, 768, c16 + (int64_t)b * Tp * 768, 768, Tp));
    TSD_TRY(g_unet_forward(m, dl, c16, T, Tp, dt, B, L, de));
    return d2h(ctx, out, de, nl * 4);
  });
}

extern "C" int tsd_decoder_forward(tsd_model* m, const float* latents, int B, int L, float* images) {
  NOTNULL(m); NOTNULL(latents); NOTNULL(images);
  if (!is_decoder_kind(m->kind)) TSD_FAIL(TSD_E_ARG, "tsd_decoder_forward: model is not a Decoder");
  if (B <= 0 || L <= 0) TSD_FAIL(TSD_E_SHAPE, "decoder: B=%d L=%d", B, L);
  tsd_ctx* ctx = m->ctx;
  return run_model(m, [&]() -> int {
    const int64_t nl = (int64_t)B * 4 * L * L, ni = (int64_t)B * 3 * 64 * L * L;
    float* dl = arena_alloc<float>(ctx, nl);
    float* di = arena_alloc<float>(ctx, ni);
    if (!dl || !di) TSD_FAIL(TSD_E_ALLOC, "workspace arena exhausted");
    TSD_TRY(h2d(ctx, dl, latents, nl * 4));
    TSD_TRY(g_decoder_forward(m, dl, B, L, di));
    return d2h(ctx, images, di, ni * 4);
  });
}

extern "C" int tsd_encoder_forward(tsd_model* m, const float* images, const float* noise, int B, int S,
                                   float* latents) {
  NOTNULL(m); NOTNULL(images); NOTNULL(noise); NOTNULL(latents);
  if (!is_encoder_kind(m->kind)) TSD_FAIL(TSD_E_ARG, "tsd_encoder_forward: model is not an Encoder");
  if (B <= 0 || S <= 0 || S % 8) TSD_FAIL(TSD_E_SHAPE, "encoder: B=%d S=%d", B, S);
  tsd_ctx* ctx = m->ctx;
  return run_model(m, [&]() -> int {
    const int L = S / 8;
    const int64_t ni = (int64_t)B * 3 * S * S, nl = (int64_t)B * 4 * L * L;
    float* di = arena_alloc<float>(ctx, ni);
    float* dn = arena_alloc<float>(ctx, nl);
    float* dl = arena_alloc<float>(ctx, nl);
    if (!di || !dn || !dl) TSD_FAIL(TSD_E_ALLOC, "workspace arena exhausted");
    TSD_TRY(h2d(ctx, di, images, ni * 4));
    TSD_TRY(h2d(ctx, dn, noise, nl * 4));
    TSD_TRY(g_encoder_forward(m, di, dn, B, S, dl));
    return d2h(ctx, latents, dl, nl * 4);
  });
}

extern "C" int tsd_clip_forward(tsd_model* m, const int32_t* tokens, int B, int T, float* context) {
  NOTNULL(m); NOTNULL(tokens); NOTNULL(context);
  if (!is_clip_kind(m->kind)) TSD_FAIL(TSD_E_ARG, "tsd_clip_forward: model is not a CLIP");
  if (B <= 0 || T <= 0 || T > 77) TSD_FAIL(TSD_E_SHAPE, "clip: B=%d T=%d (1..77 tokens)", B, T);
  tsd_ctx* ctx = m->ctx;
  std::vector<int32_t> padded((size_t)B * 77, 0);  // clip.mojo:91-93: a zero row of 77 ids, the prompt's ids in front
  for (int b = 0; b < B; b++)
    for (int t = 0; t < T; t++) padded[(size_t)b * 77 + t] = tokens[(size_t)b * T + t];
  return run_model(m, [&]() -> int {
    const int64_t no = (int64_t)B * 77 * 768;
    int* dt = arena_alloc<int>(ctx, (int64_t)B * 77);
    float* dout = arena_alloc<float>(ctx, no);
    if (!dt || !dout) TSD_FAIL(TSD_E_ALLOC, "workspace arena exhausted");
    TSD_TRY(h2d(ctx, dt, padded.data(), padded.size() * sizeof(int32_t)));
    TSD_TRY(g_clip_forward(m, dt, B, dout));
    return d2h(ctx, context, dout, no * 4);
  });
}

// ---- device-resident denoise session ---------------------------------------------------------------
struct tsd_session {
  tsd_model* unet = nullptr;
  tsd_model* dec = nullptr;
  tsd_ctx* ctx = nullptr;
  int B = 0, L = 0, T = 0, Tp = 0, cfg = 0;
  float cfg_scale = 7.5f;
  // persistent device state (own allocation, not the arena)
  char* state = nullptr;
  float* latents = nullptr;   // [B,4,L,L] fp32 CHW (the reference's own layout)
  float* lat2 = nullptr;      // [2B,4,L,L] duplicated latents for the CFG batch
  half_t* ctx16 = nullptr;    // [B or 2B][Tp][768]
  float* eps = nullptr;       // [B or 2B,4,L,L]
  float* tdev = nullptr;      // [16] timestep per sample
  float* temb = nullptr;      // [16][320]
  float* noise = nullptr;     // [nsteps,B,4,L,L] or null
  float* images = nullptr;    // [B,3,8L,8L]
  size_t noise_cap = 0;
  // schedule (sampler.mojo:15-44)
  int n_train = 1000, n_infer = 50, start = 0;
  std::vector<float> alphas_cumprod;
  std::vector<int> timesteps;
  bool uploaded = false, has_noise = false;
  size_t plan_unet = 0, plan_dec = 0;
  unsigned opt_gen = 0;  // generation of the context's options the workspace was sized for (upload)
  // Latched when the host scan of a download found inf / NaN in THIS session's latents: the context's counter is cleared once reported,
  // the latents stay what they are - every later step, decode and download of this session fails with TSD_E_NONFINITE until upload()
  // replaces the state.  The context-wide counter alone does not latch it: it also counts other sessions and
  // models of the context (bench.py's scaled "peaked" model copy); its code is still returned at the synchronisation point that sees it.
  bool poisoned = false;
  bool decoded = false;  // decode() has run since the last upload(): images are defined
};

// inf / NaN scan of a downloaded tensor (exponent bits all ones); the buffers at this boundary are 0.5 - 25 MB
static bool host_all_finite(const float* p, size_t n) {
  unsigned bad = 0;
  for (size_t i = 0; i < n; i++) {
    unsigned u;
    memcpy(&u, p + i, 4);
    bad |= ((u & 0x7f800000u) == 0x7f800000u);
  }
  return bad == 0;
}
#define SESSION_NOT_POISONED(s)                                                                                              \
  do {                                                                                                                       \
    if ((s)->poisoned)                                                                                                       \
      TSD_FAIL(TSD_E_NONFINITE, "session: inf / NaN was found in this session's latents or images (reported at an earlier " \
               "download); its state is unusable until upload() replaces it");                                              \
  } while (0)

static void build_schedule(tsd_session* s) {
  // betas = linspace(sqrt(b0), sqrt(b1), N)^2 ; alphas_cumprod = cumprod(1 - betas)   (sampler.mojo:28-32), fp32
  const int N = s->n_train;
  s->alphas_cumprod.resize(N);
  const float b0 = sqrtf(0.00085f), b1 = sqrtf(0.0120f);
  float prod = 1.f;
  for (int i = 0; i < N; i++) {
    const float step = N > 1 ? (b1 - b0) / (float)(N - 1) : 0.f;  // numpy.linspace order: i*step + start
    const float v = (float)i * step + b0;
    const float beta = v * v;
    prod *= (1.f - beta);
    s->alphas_cumprod[i] = prod;
  }
  // timesteps = round(arange(n)[::-1] * (N // n))  (sampler.mojo:40-43), then drop `start` (set_strength, App.A D21)
  s->timesteps.clear();
  const int ratio = N / s->n_infer;
  for (int i = s->n_infer - 1; i >= 0; i--) s->timesteps.push_back(i * ratio);
  if (s->start > 0) s->timesteps.erase(s->timesteps.begin(), s->timesteps.begin() + std::min<size_t>(s->start, s->timesteps.size()));
}

extern "C" int tsd_session_create(tsd_model* diffusion, tsd_model* decoder, int B, int L, int T, int cfg,
                                  tsd_session** out) {
  NOTNULL(diffusion); NOTNULL(out);
  if (!is_diffusion_kind(diffusion->kind)) TSD_FAIL(TSD_E_ARG, "session: first model must be a Diffusion");
  if (decoder && (!is_decoder_kind(decoder->kind) || decoder->ctx != diffusion->ctx))
    TSD_FAIL(TSD_E_ARG, "session: decoder must be a Decoder on the same context");
  const int Bu = cfg ? 2 * B : B;
  if (B <= 0 || Bu > 16 || L <= 0 || L % 8 || T <= 0) TSD_FAIL(TSD_E_SHAPE, "session: B=%d (UNet batch %d <= 16) L=%d T=%d", B, Bu, L, T);
  tsd_ctx* ctx = diffusion->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  tsd_session* s = new tsd_session();
  s->unet = diffusion; s->dec = decoder; s->ctx = ctx;
  s->B = B; s->L = L; s->T = T; s->Tp = round_up(T, 8); s->cfg = cfg ? 1 : 0;
  const size_t nl = (size_t)B * 4 * L * L;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = (off + bytes + 255) & ~size_t(255); return o; };
  const size_t o_lat = carve(nl * 4), o_lat2 = carve(2 * nl * 4), o_ctx = carve((size_t)Bu * s->Tp * 768 * 2),
               o_eps = carve((size_t)Bu * 4 * L * L * 4), o_t = carve(16 * 4), o_te = carve(16 * 320 * 4),
               o_img = carve(decoder ? (size_t)B * 3 * 64 * L * L * 4 : 0);
  hipError_t e = hipMalloc((void**)&s->state, off);
  if (e != hipSuccess) { delete s; TSD_FAIL(TSD_E_ALLOC, "session: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e)); }
  if (ctx->opt.debug_poison >= 0 && (ctx->opt.debug_poison_what & 4)) hipMemsetAsync(s->state, ctx->opt.debug_poison & 255, off, ctx->stream);
  s->latents = (float*)(s->state + o_lat); s->lat2 = (float*)(s->state + o_lat2);
  s->ctx16 = (half_t*)(s->state + o_ctx); s->eps = (float*)(s->state + o_eps);
  s->tdev = (float*)(s->state + o_t); s->temb = (float*)(s->state + o_te);
  s->images = decoder ? (float*)(s->state + o_img) : nullptr;
  build_schedule(s);
  *out = s;
  return TSD_OK;
}

extern "C" int tsd_session_destroy(tsd_session* s) {
  if (!s) return TSD_OK;
  hipSetDevice(s->ctx->device);
  hipStreamSynchronize(s->ctx->stream);
  if (s->state) hipFree(s->state);
  if (s->noise) hipFree(s->noise);
  delete s;
  return TSD_OK;
}

extern "C" int tsd_session_set_schedule(tsd_session* s, int num_training_steps, int num_inference_steps,
                                        int start_step) {
  NOTNULL(s);
  if (num_training_steps <= 0 || num_inference_steps <= 0 || num_inference_steps > num_training_steps ||
      start_step < 0 || start_step >= num_inference_steps)
    TSD_FAIL(TSD_E_ARG, "schedule: train=%d infer=%d start=%d", num_training_steps, num_inference_steps, start_step);
  s->n_train = num_training_steps; s->n_infer = num_inference_steps; s->start = start_step;
  build_schedule(s);
  // the device noise buffer and the workspace plan were sized for the OLD schedule: a new upload() is required
  s->uploaded = false; s->has_noise = false;
  return TSD_OK;
}
extern "C" int tsd_session_num_steps(tsd_session* s) { return s ? (int)s->timesteps.size() : TSD_E_ARG; }
extern "C" int tsd_session_timestep(tsd_session* s, int i) {
  if (!s || i < 0 || i >= (int)s->timesteps.size()) return TSD_E_ARG;
  return s->timesteps[i];
}

extern "C" int tsd_session_upload(tsd_session* s, const float* latents, const float* context,
                                  const float* uncond_context, const float* noise, float cfg_scale) {
  NOTNULL(s); NOTNULL(latents); NOTNULL(context);
  if (s->cfg && !uncond_context) TSD_FAIL(TSD_E_ARG, "session: CFG session needs uncond_context");
  tsd_ctx* ctx = s->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  const int B = s->B, L = s->L, T = s->T, Tp = s->Tp;
  const size_t nl = (size_t)B * 4 * L * L;
  s->cfg_scale = cfg_scale;
  HIP_TRY(hipMemcpyAsync(s->latents, latents, nl * 4, hipMemcpyHostToDevice, ctx->stream));
  // context -> fp16 [Bu][Tp][768], zero padded rows (cond first, then uncond: pipeline.mojo:49)
  TSD_TRY(ctx_reserve_staging(ctx, (size_t)B * T * 768 * 4));
  for (int part = 0; part < (s->cfg ? 2 : 1); part++) {
    const float* src = part == 0 ? context : uncond_context;
    HIP_TRY(hipMemcpyAsync(ctx->staging, src, (size_t)B * T * 768 * 4, hipMemcpyHostToDevice, ctx->stream));
    for (int b = 0; b < B; b++)
      TSD_TRY(launch_f32_to_f16_rows(ctx, (const float*)ctx->staging + (size_t)b * T * 768, T, 768,
                                     s->ctx16 + ((size_t)part * B + b) * Tp * 768, 768, Tp));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  s->has_noise = noise != nullptr;
  if (noise) {
    const size_t bytes = s->timesteps.size() * nl * 4;
    if (bytes > s->noise_cap) {
      if (s->noise) HIP_TRY(hipFree(s->noise));
      s->noise = nullptr; s->noise_cap = 0;
      hipError_t e = hipMalloc((void**)&s->noise, bytes);
      if (e != hipSuccess) TSD_FAIL(TSD_E_ALLOC, "session: noise hipMalloc(%zu) failed", bytes);
      s->noise_cap = bytes;
    }
    HIP_TRY(hipMemcpyAsync(s->noise, noise, bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  // size the workspace once for this session's shapes (no allocation inside the timed loop)
  TSD_TRY(model_check_ready(s->unet));
  if (s->dec) TSD_TRY(model_check_ready(s->dec));
  Arena& a = ctx->arena;
  const int Bu = s->cfg ? 2 * B : B;
  a.planning = true; a.top = 0; a.peak = 0;
  int r = g_unet_forward(s->unet, s->lat2, s->ctx16, T, Tp, s->temb, Bu, L, s->eps);
  size_t need = a.peak;
  if (r == TSD_OK && s->dec) {
    a.top = 0; a.peak = 0;
    r = g_decoder_forward(s->dec, s->latents, B, L, s->images);
    need = std::max(need, a.peak);
  }
  a.planning = false; a.top = 0; a.peak = 0;
  if (r != TSD_OK) return r;
  TSD_TRY(ctx_reserve_arena(ctx, need));
  s->opt_gen = ctx->opt.gen;
  s->uploaded = true;
  s->poisoned = false;
  s->decoded = false;
  return TSD_OK;
}

// scalar coefficients of `DDPMSampler.step` sampler.mojo:81-98 and `get_variance` :53-65 (fp32 like the reference)
static void ddpm_coeffs(const tsd_session* s, int t, float* sa, float* sb, float* c_x0, float* c_xt, float* sigma) {
  const int prev = t - s->n_train / s->n_infer;
  const float a_t = s->alphas_cumprod[t];
  const float a_prev = prev >= 0 ? s->alphas_cumprod[prev] : 1.f;
  const float b_t = 1.f - a_t, b_prev = 1.f - a_prev;
  const float cur_a = a_t / a_prev, cur_b = 1.f - cur_a;
  *sa = sqrtf(a_t); *sb = sqrtf(b_t);
  *c_x0 = sqrtf(a_prev) * cur_b / b_t;
  *c_xt = sqrtf(cur_a) * b_prev / b_t;
  float var = b_prev / b_t * cur_b;
  if (var < 1e-20f) var = 1e-20f;
  *sigma = t > 0 ? sqrtf(var) : 0.f;
}

extern "C" int tsd_session_step(tsd_session* s, int i) {
  NOTNULL(s);
  if (!s->uploaded) TSD_FAIL(TSD_E_STATE, "session: upload() before step()");
  SESSION_NOT_POISONED(s);
  if (s->opt_gen != s->ctx->opt.gen) TSD_FAIL(TSD_E_STATE, "session: a tsd_debug_set_* call changed this context's options after upload() sized the workspace; upload() again");
  if (i < 0 || i >= (int)s->timesteps.size()) TSD_FAIL(TSD_E_ARG, "session: step %d out of range", i);
  tsd_ctx* ctx = s->ctx;
  const int B = s->B, L = s->L, Bu = s->cfg ? 2 * B : B;
  const size_t nl = (size_t)B * 4 * L * L;
  const int t = s->timesteps[i];
  // time embedding on the device (get_time_embedding, pipeline.mojo:89): same t for every sample
  TSD_TRY(launch_time_embedding(ctx, nullptr, (float)t, Bu, s->temb));
  const float* lat_in = s->latents;
  if (s->cfg) {  // model_input for both passes is the same latents (pipeline.mojo:107-108)
    HIP_TRY(hipMemcpyAsync(s->lat2, s->latents, nl * 4, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(s->lat2 + nl, s->latents, nl * 4, hipMemcpyDeviceToDevice, ctx->stream));
    lat_in = s->lat2;
  }
  ctx->arena.top = 0;
  // eps stays in the output convolution's layout ([B][L*L][4]); the DDPM update reads it as such (no conversion launch)
  const bool eps_nhwc = s->unet->unet.final_conv.Opad == 4;  // g_unet_forward's condition for writing that layout
  TSD_TRY(g_unet_forward(s->unet, lat_in, s->ctx16, s->T, s->Tp, s->temb, Bu, L, s->eps, eps_nhwc));
  ctx->arena.top = 0;
  float sa, sb, c_x0, c_xt, sigma;
  ddpm_coeffs(s, t, &sa, &sb, &c_x0, &c_xt, &sigma);
  const float* nz = (s->has_noise && t > 0) ? s->noise + (size_t)i * nl : nullptr;
  return launch_ddpm_step(ctx, s->latents, s->eps, s->cfg ? s->eps + nl : nullptr, s->cfg_scale, nz, (int64_t)nl, sa, sb,
                          c_x0, c_xt, sigma, eps_nhwc ? L * L : 0);
}

extern "C" int tsd_session_add_noise(tsd_session* s, int i, const float* noise) {
  NOTNULL(s); NOTNULL(noise);
  if (i < 0 || i >= (int)s->timesteps.size()) TSD_FAIL(TSD_E_ARG, "session: step %d out of range", i);
  tsd_ctx* ctx = s->ctx;
  const size_t nl = (size_t)s->B * 4 * s->L * s->L;
  TSD_TRY(ctx_reserve_staging(ctx, nl * 4));
  HIP_TRY(hipMemcpyAsync(ctx->staging, noise, nl * 4, hipMemcpyHostToDevice, ctx->stream));
  const float a = s->alphas_cumprod[s->timesteps[i]];
  TSD_TRY(launch_add_noise(ctx, s->latents, (const float*)ctx->staging, (int64_t)nl, sqrtf(a), sqrtf(1.f - a)));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return TSD_OK;
}

extern "C" int tsd_session_decode(tsd_session* s) {
  NOTNULL(s);
  if (!s->dec) TSD_FAIL(TSD_E_STATE, "session: created without a decoder");
  if (!s->uploaded) TSD_FAIL(TSD_E_STATE, "session: upload() before decode()");
  SESSION_NOT_POISONED(s);
  if (s->opt_gen != s->ctx->opt.gen) TSD_FAIL(TSD_E_STATE, "session: a tsd_debug_set_* call changed this context's options after upload() sized the workspace; upload() again");
  s->ctx->arena.top = 0;
  int r = g_decoder_forward(s->dec, s->latents, s->B, s->L, s->images);
  s->ctx->arena.top = 0;
  if (r == TSD_OK) s->decoded = true;
  return r;
}

extern "C" int tsd_session_download_latents(tsd_session* s, float* latents) {
  NOTNULL(s); NOTNULL(latents);
  if (!s->uploaded) TSD_FAIL(TSD_E_STATE, "session: upload() before download_latents() (the state is not defined yet)");
  const size_t n = (size_t)s->B * 4 * s->L * s->L;
  HIP_TRY(hipMemcpyAsync(latents, s->latents, n * 4, hipMemcpyDeviceToHost, s->ctx->stream));
  HIP_TRY(hipStreamSynchronize(s->ctx->stream));
  const int r = ctx_check_status(s->ctx);
  // what leaves the device HERE is checked itself: the context's count may have been reported (and cleared) at another model's
  // synchronisation point, and it may be another session's - only this session's own buffer latches the poison
  if (!host_all_finite(latents, n)) s->poisoned = true;
  if (r != TSD_OK) return r;
  SESSION_NOT_POISONED(s);
  return TSD_OK;
}

extern "C" int tsd_session_download_images(tsd_session* s, int rescale_0_255, float* images) {
  NOTNULL(s); NOTNULL(images);
  if (!s->dec) TSD_FAIL(TSD_E_STATE, "session: created without a decoder");
  if (!s->decoded) TSD_FAIL(TSD_E_STATE, "session: decode() before download_images() (no image has been computed since upload())");
  tsd_ctx* ctx = s->ctx;
  const int64_t n = (int64_t)s->B * 3 * 64 * s->L * s->L;
  const float* src = s->images;
  if (rescale_0_255) {  // pipeline.mojo:127
    TSD_TRY(ctx_reserve_staging(ctx, (size_t)n * 4));
    TSD_TRY(launch_unary_f32(ctx, 2, s->images, n, (float*)ctx->staging));
    src = (const float*)ctx->staging;
  }
  HIP_TRY(hipMemcpyAsync(images, src, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  const int r = ctx_check_status(ctx);
  if (r != TSD_OK) return r;
  SESSION_NOT_POISONED(s);
  // a non-finite IMAGE is reported, it does not poison latents that may be fine (decode again after the cause is removed)
  if (!host_all_finite(images, (size_t)n)) TSD_FAIL(TSD_E_NONFINITE, "session: inf / NaN in the decoded images");
  return TSD_OK;
}

// ---- RCCL weight broadcast over xGMI ---------------------------------------
// librccl is resolved lazily with dlopen so single-GPU users never load it (and a process that already
// has torch's RCCL loaded reuses that copy: same SONAME).
namespace {
typedef struct { char internal[128]; } nccl_uid;
typedef int (*fn_get_uid)(nccl_uid*);
typedef int (*fn_init_rank)(void**, int, nccl_uid, int);
typedef int (*fn_bcast)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(void*, int*);
struct Rccl {
  void* lib = nullptr;
  fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_bcast bcast = nullptr; fn_destroy destroy = nullptr;
  fn_errstr errstr = nullptr;
  fn_count count = nullptr;
};
Rccl g_rccl;
int rccl_load() {
  if (g_rccl.lib) return TSD_OK;
  // A process must use ONE HIP runtime: torch wheels bundle their own libamdhip64 + librccl, and an RCCL built against the
  // other runtime fails in ncclCommInitRank ("unhandled cuda error": round-1 log, libtsd loaded /opt/rocm's runtime first
  // and a later `import torch` mapped a second one).  So: first take an RCCL that is ALREADY mapped (RTLD_NOLOAD; it belongs
  // to the runtime the process is using - with torch imported first libtsd itself binds to torch's libamdhip64, same
  // SONAME), and only otherwise load the system one.  scripts/diag_rccl_coresident.py shows the three cases.
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) {
    g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (g_rccl.lib) break;
  }
  for (int i = 0; i < 3 && !g_rccl.lib; i++) {
    const char* order[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    g_rccl.lib = dlopen(order[i], RTLD_NOW | RTLD_GLOBAL);
  }
  if (!g_rccl.lib) TSD_FAIL(TSD_E_RCCL, "cannot dlopen librccl: %s", dlerror());
  g_rccl.get_uid = (fn_get_uid)dlsym(g_rccl.lib, "ncclGetUniqueId");
  g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.lib, "ncclCommInitRank");
  g_rccl.bcast = (fn_bcast)dlsym(g_rccl.lib, "ncclBroadcast");
  g_rccl.destroy = (fn_destroy)dlsym(g_rccl.lib, "ncclCommDestroy");
  g_rccl.errstr = (fn_errstr)dlsym(g_rccl.lib, "ncclGetErrorString");
  g_rccl.count = (fn_count)dlsym(g_rccl.lib, "ncclCommCount");
  if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.bcast || !g_rccl.destroy)
    TSD_FAIL(TSD_E_RCCL, "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclBroadcast");
  return TSD_OK;
}
#define RCCL_TRY(expr)                                                                                     \
  do {                                                                                                     \
    int r__ = (expr);                                                                                      \
    if (r__ != 0) TSD_FAIL(TSD_E_RCCL, "%s failed: %s", #expr, g_rccl.errstr ? g_rccl.errstr(r__) : "?"); \
  } while (0)
}  // namespace

extern "C" int tsd_dist_unique_id(void* id128) {
  NOTNULL(id128);
  TSD_TRY(rccl_load());
  nccl_uid id;
  RCCL_TRY(g_rccl.get_uid(&id));
  memcpy(id128, &id, 128);
  return TSD_OK;
}

extern "C" int tsd_dist_init(tsd_ctx* ctx, int rank, int nranks, const void* id128) {
  NOTNULL(ctx); NOTNULL(id128);
  if (nranks <= 0 || rank < 0 || rank >= nranks) TSD_FAIL(TSD_E_ARG, "dist: rank %d / %d", rank, nranks);
  TSD_TRY(rccl_load());
  HIP_TRY(hipSetDevice(ctx->device));
  nccl_uid id;
  memcpy(&id, id128, 128);
  void* comm = nullptr;
  RCCL_TRY(g_rccl.init_rank(&comm, nranks, id, rank));
  ctx->rccl_comm = comm; ctx->rank = rank; ctx->nranks = nranks;
  return TSD_OK;
}

extern "C" int tsd_dist_broadcast_weights(tsd_model* m, int root) {
  NOTNULL(m);
  tsd_ctx* ctx = m->ctx;
  if (!ctx->rccl_comm) TSD_FAIL(TSD_E_STATE, "dist: tsd_dist_init was not called on this context");
  HIP_TRY(hipSetDevice(ctx->device));
  // one broadcast of the whole packed blob (fp16 weights + fp32 biases), ncclInt8 = 0
  RCCL_TRY(g_rccl.bcast(m->blob, m->blob, m->blob_bytes, 0 /*ncclInt8*/, root, ctx->rccl_comm, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return tsd_model_mark_loaded(m);
}

extern "C" int tsd_dist_comm_count(tsd_ctx* ctx, int* nranks) {
  NOTNULL(ctx); NOTNULL(nranks);
  if (!ctx->rccl_comm) TSD_FAIL(TSD_E_STATE, "dist: tsd_dist_init was not called on this context");
  if (!g_rccl.count) TSD_FAIL(TSD_E_RCCL, "librccl has no ncclCommCount");
  RCCL_TRY(g_rccl.count(ctx->rccl_comm, nranks));
  return TSD_OK;
}

extern "C" int tsd_dist_finalize(tsd_ctx* ctx) {
  NOTNULL(ctx);
  if (ctx->rccl_comm && g_rccl.destroy) g_rccl.destroy(ctx->rccl_comm);
  ctx->rccl_comm = nullptr;
  return TSD_OK;
}
