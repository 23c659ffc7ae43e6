// model.cpp - device-resident packed weights: one blob per model (what RCCL broadcasts),
// per-parameter upload/pack, synthetic counter-RNG init, and the resolved weight trees.
#include <string.h>

#include "model.h"

bool attn_tail_weights_ok(const AttnW& w) {
  const int C = w.C;
  const bool plain = !w.ln[1].w && !w.ln[1].b && !w.ln[2].w && !w.ln[2].b && !w.gelu_erf;
  return plain && C == 320 && w.n_head == 8 && w.n_embed == 40 && w.sa_out.b && w.ca_out.b && w.geglu1.b && w.geglu2.b &&
         w.conv_out.b && !w.ca_q.b && w.sa_out.Kpad == C && w.ca_q.Kpad == C && w.ca_out.Kpad == C && w.geglu1.Kpad == C &&
         w.geglu2.Kpad == 4 * C && w.conv_out.Ipad == C && w.conv_out.k == 1;
}

bool attn_head_weights_ok(const AttnW& w) {
  const int C = w.C;
  return C == 320 && w.n_head == 8 && w.n_embed == 40 && !w.gn.w && !w.gn.b && !w.ln[0].w && !w.ln[0].b && w.conv_in.b &&
         w.conv_in.k == 1 && w.conv_in.Ipad == C && w.conv_in.Opad == C && !w.sa_in.b && w.sa_in.Kpad == C && w.sa_in.N == 3 * C;
}

// parameters changed: derived buffers are stale until the next model_check_ready()
static void model_invalidate_derived(tsd_model* m) {
  m->ready = false;
  for (auto& a : m->unet.attn) { a.tail_stream = nullptr; a.head_stream = nullptr; a.fold_w = nullptr; a.fold_w_tm = nullptr; a.fold_b = nullptr; }
  m->unet.conv_in_im2col = nullptr;
  m->vae.conv_in_im2col = nullptr;
  for (auto& r : m->unet.res) { r.conv1.w_tm = nullptr; r.conv2.w_tm = nullptr; }
  m->unet.conv4.w_tm = nullptr; m->unet.conv7.w_tm = nullptr; m->unet.kproj_all.w_tm = nullptr;
  for (auto& cv : m->unet.conv) cv.w_tm = nullptr;
  for (auto& a : m->unet.attn) {
    for (LinW* l : {&a.sa_in, &a.sa_out, &a.ca_q, &a.ca_out, &a.geglu1, &a.geglu2}) l->w_tm = nullptr;
    a.conv_in.w_tm = nullptr; a.conv_out.w_tm = nullptr;
  }
}

static size_t packed_bytes(const ParamSpec& p) {
  if (!p.used) return 0;
  switch (p.kind) {
    case P_CONV_W: return (size_t)p.Opad * p.shape[2] * p.shape[3] * p.Kpad * sizeof(half_t);
    case P_CONV_B: return (size_t)p.Opad * sizeof(float);
    case P_LIN_W: return (size_t)p.shape[0] * p.Kpad * sizeof(half_t);
    default: return (size_t)p.shape[0] * sizeof(float);
  }
}

extern "C" int tsd_model_create(tsd_ctx* ctx, int kind, tsd_model** out) {
  if (!ctx || !out) TSD_FAIL(TSD_E_ARG, "tsd_model_create: NULL argument");
  if (kind < TSD_MODEL_DIFFUSION || kind > TSD_MODEL_KIND_MAX) TSD_FAIL(TSD_E_ARG, "bad model kind %d", kind);
  tsd_model* m = new tsd_model();
  m->ctx = ctx;
  m->kind = kind;
  m->params = build_param_specs(kind);
  // blob layout: time-projection weights (region 1) and biases (region 2) are laid out contiguously in
  // layer order so the nine Linear(1280,C) become ONE [6720][1280] GEMV (SURVEY.md App.D K8).
  size_t off = 0;
  for (auto& p : m->params) p.bytes = packed_bytes(p);
  {  // regions 1/2 densely packed (no gaps between the nine tensors), everything else 256-B aligned
    size_t o = 0;
    for (auto& p : m->params) if (p.region == 1) { p.off = o; o += p.bytes; }
    o = (o + 255) & ~size_t(255);
    for (auto& p : m->params) if (p.region == 2) { p.off = o; o += p.bytes; }
    o = (o + 255) & ~size_t(255);
    for (auto& p : m->params) if (p.region == 3) { p.off = o; o += p.bytes; }
    o = (o + 255) & ~size_t(255);
    for (auto& p : m->params) if (p.region == 4) { p.off = o; o += p.bytes; }
    o = (o + 255) & ~size_t(255);
    for (auto& p : m->params) if (p.region == 0) { p.off = o; o += p.bytes; o = (o + 255) & ~size_t(255); }
    off = o;
  }
  m->blob_bytes = off;
  for (size_t i = 0; i < m->params.size(); i++) m->index[m->params[i].name] = (int)i;
  m->loaded.assign(m->params.size(), 0);
  HIP_TRY(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc((void**)&m->blob, m->blob_bytes);
  if (e != hipSuccess) {
    delete m;
    TSD_FAIL(TSD_E_ALLOC, "weights: hipMalloc(%zu) failed: %s", off, hipGetErrorString(e));
  }
  HIP_TRY(hipMemsetAsync(m->blob, 0, m->blob_bytes, ctx->stream));
  TSD_TRY(model_resolve(m));
  *out = m;
  return TSD_OK;
}

extern "C" int tsd_model_destroy(tsd_model* m) {
  if (!m) return TSD_OK;
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  if (m->blob) hipFree(m->blob);
  if (m->derived) hipFree(m->derived);
  delete m;
  return TSD_OK;
}

// pack parameter `i` from a device fp32 tensor in the reference layout
static int pack_param(tsd_model* m, int i, const float* dev_src) {
  tsd_ctx* ctx = m->ctx;
  const ParamSpec& p = m->params[i];
  if (!p.used) return TSD_OK;
  char* dst = m->blob + p.off;
  switch (p.kind) {
    case P_CONV_W:
      return launch_pack_conv(ctx, dev_src, (int)p.shape[0], (int)p.shape[1], (int)p.shape[2], (half_t*)dst, p.Opad,
                              p.Kpad);
    case P_CONV_B: return launch_pack_bias(ctx, dev_src, (int)p.shape[0], (float*)dst, p.Opad, 0);
    case P_LIN_W:
      return launch_pack_linear(ctx, dev_src, (int)p.shape[0], (int)p.shape[1], (half_t*)dst, p.Kpad, p.interleave);
    default: return launch_pack_bias(ctx, dev_src, (int)p.shape[0], (float*)dst, (int)p.shape[0], p.interleave);
  }
}

extern "C" int tsd_model_set_param(tsd_model* m, int index, const float* data, int64_t numel) {
  if (!m || !data) TSD_FAIL(TSD_E_ARG, "tsd_model_set_param: NULL argument");
  if (index < 0 || index >= (int)m->params.size()) TSD_FAIL(TSD_E_ARG, "param index %d out of range", index);
  const ParamSpec& p = m->params[index];
  if (numel != p.numel())
    TSD_FAIL(TSD_E_SHAPE, "param %s: got %lld elements, expected %lld", p.name.c_str(), (long long)numel,
             (long long)p.numel());
  m->loaded[index] = 1;
  if (!p.used) return TSD_OK;  // allocated by the reference, never read by its forward
  tsd_ctx* ctx = m->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  TSD_TRY(ctx_reserve_staging(ctx, (size_t)numel * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(ctx->staging, data, (size_t)numel * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  TSD_TRY(pack_param(m, index, (const float*)ctx->staging));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  model_invalidate_derived(m);
  return TSD_OK;
}

extern "C" int tsd_model_init_random(tsd_model* m, uint64_t seed) {
  if (!m) TSD_FAIL(TSD_E_ARG, "tsd_model_init_random: NULL model");
  tsd_ctx* ctx = m->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  int64_t maxn = 0;
  for (auto& p : m->params) if (p.used) maxn = std::max(maxn, p.numel());
  TSD_TRY(ctx_reserve_staging(ctx, (size_t)maxn * sizeof(float)));
  for (size_t i = 0; i < m->params.size(); i++) {
    const ParamSpec& p = m->params[i];
    m->loaded[i] = 1;
    if (!p.used) continue;
    float* tmp = (float*)ctx->staging;
    if (p.bound == 0.f) HIP_TRY(hipMemsetAsync(tmp, 0, (size_t)p.numel() * sizeof(float), ctx->stream));
    else TSD_TRY(launch_fill_uniform(ctx, tmp, p.numel(), seed, (uint64_t)m->kind * 4096 + i, p.bound));
    if (p.offset != 0.f) TSD_TRY(launch_add_const_f32(ctx, tmp, p.numel(), p.offset));
    TSD_TRY(pack_param(m, (int)i, tmp));
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  model_invalidate_derived(m);
  return TSD_OK;
}

extern "C" int tsd_model_packed_blob(tsd_model* m, void** device_ptr, size_t* bytes) {
  if (!m || !device_ptr || !bytes) TSD_FAIL(TSD_E_ARG, "tsd_model_packed_blob: NULL argument");
  *device_ptr = m->blob;
  *bytes = m->blob_bytes;
  return TSD_OK;
}

extern "C" int tsd_model_mark_loaded(tsd_model* m) {
  if (!m) TSD_FAIL(TSD_E_ARG, "NULL model");
  std::fill(m->loaded.begin(), m->loaded.end(), 1);
  model_invalidate_derived(m);
  return TSD_OK;
}

// derived device buffers (not part of the broadcast blob: every rank rebuilds them from the packed weights)
static int model_build_derived(tsd_model* m) {
  tsd_ctx* ctx = m->ctx;
  if (is_decoder_kind(m->kind) || is_encoder_kind(m->kind)) {  // first convolution of the VAE halves: im2col weights
    if (m->vae.conv.empty()) return TSD_OK;
    const ConvW& c0 = m->vae.conv[0];
    if (!(c0.w && c0.k == 3 && c0.I > 0 && 9 * c0.I <= 64 && c0.Ipad == 64)) return TSD_OK;
    const size_t need = (size_t)c0.Opad * 64 * sizeof(half_t);
    HIP_TRY(hipSetDevice(ctx->device));
    if (m->derived_bytes < need) {
      if (m->derived) HIP_TRY(hipFree(m->derived));
      m->derived = nullptr; m->derived_bytes = 0;
      hipError_t e = hipMalloc((void**)&m->derived, need);
      if (e != hipSuccess) TSD_FAIL(TSD_E_ALLOC, "derived weights: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
      m->derived_bytes = need;
      if (ctx->opt.debug_poison >= 0 && (ctx->opt.debug_poison_what & 2)) HIP_TRY(hipMemsetAsync(m->derived, ctx->opt.debug_poison & 255, need, ctx->stream));
    }
    const bool wp = ctx->arena.planning;
    ctx->arena.planning = false;
    const int r = launch_pack_im2col_w(ctx, c0.w, c0.Opad, c0.Ipad, c0.I, (half_t*)m->derived);
    ctx->arena.planning = wp;
    if (r == TSD_OK) m->vae.conv_in_im2col = (const half_t*)m->derived;
    return r;
  }
  if (!is_diffusion_kind(m->kind)) return TSD_OK;
  std::vector<AttnW*> el;
  for (auto& a : m->unet.attn) if (a.C && (attn_tail_weights_ok(a) || attn_head_weights_ok(a))) el.push_back(&a);
  // input convolution (4 latent channels): [Opad][64] im2col weights
  const ConvW& cin = is_full_unet_kind(m->kind) ? (m->unet.conv.empty() ? m->unet.conv1 : m->unet.conv[0]) : m->unet.conv1;
  const bool cin_ok = cin.w && cin.k == 3 && cin.I > 0 && 9 * cin.I <= 64 && cin.Ipad == 64;
  const size_t cin_b = cin_ok ? ((size_t)cin.Opad * 64 * sizeof(half_t) + 255) & ~size_t(255) : 0;
  // weight-heavy 3x3 convs (>= TSD_CONV_W_TM MiB of weights; 0 = off): K-tile-major copies
  const int tm_mib = ctx->opt.conv_w_tm_mib;  // measured: +0.3 % headline, +0.6 % full-size UNet (profiles/r03_conv_w_tile_major_ab.txt)
  // linear layers / 1x1 convs: TSD_LIN_W_TM = 0 turns them off, TSD_LIN_W_TM_KIB sets the size threshold (default 1 MiB; with 512 KiB the
  // 800-KB 640 x 640 projections of the 32x32 level join in: measured equal, 205.65 vs 205.60 steps/s)
  const int tml_kib = ctx->opt.lin_w_tm_kib;
  const size_t tml_bytes = (size_t)(tml_kib > 0 ? tml_kib : 1024) << 10;
  std::vector<ConvW*> tm;
  size_t tm_b = 0;
  if (tm_mib > 0)
    for (auto& r : m->unet.res)
      for (ConvW* c : {&r.conv1, &r.conv2})
        if (c->w && c->k == 3 && (size_t)c->Opad * 9 * c->Ipad * 2 >= (size_t)tm_mib << 20) { tm.push_back(c); tm_b += (((size_t)c->Opad * 9 * c->Ipad * 2) + 255) & ~size_t(255); }
  if (tm_mib > 0) {  // downsampling / upsampling convs outside the residual blocks
    for (ConvW* c : {&m->unet.conv4, &m->unet.conv7})
      if (c->w && c->k == 3 && (size_t)c->Opad * 9 * c->Ipad * 2 >= (size_t)tm_mib << 20) { tm.push_back(c); tm_b += (((size_t)c->Opad * 9 * c->Ipad * 2) + 255) & ~size_t(255); }
    for (auto& cv : m->unet.conv)
      if (cv.w && cv.k == 3 && (size_t)cv.Opad * 9 * cv.Ipad * 2 >= (size_t)tm_mib << 20) { tm.push_back(&cv); tm_b += (((size_t)cv.Opad * 9 * cv.Ipad * 2) + 255) & ~size_t(255); }
  }
  if (tm_mib > 0 && ctx->opt.lin_w_tm != 0)
    for (auto& a : m->unet.attn)
      if (a.C && !(attn_tail_weights_ok(a) && attn_head_weights_ok(a)))
        for (ConvW* c : {&a.conv_in, &a.conv_out})
          if (c->w && c->k == 1 && c->Ipad % 64 == 0 && (size_t)c->Opad * c->Ipad * 2 >= tml_bytes) { tm.push_back(c); tm_b += (((size_t)c->Opad * c->Ipad * 2) + 255) & ~size_t(255); }
  // ... and of the attention blocks' linear layers (TSD_LIN_W_TM MiB; the 64x64-level blocks read theirs through the fused kernels' streams)
  const int tml_mib = ctx->opt.lin_w_tm;  // measured: +0.4 % headline, +0.6 % full-size UNet (profiles/r03_lin_w_tile_major_ab.txt)
  std::vector<LinW*> tml;
  if (tml_mib > 0)
    for (auto& a : m->unet.attn)
      if (a.C && !(attn_tail_weights_ok(a) && attn_head_weights_ok(a)))
        for (LinW* l : {&a.sa_in, &a.sa_out, &a.ca_q, &a.ca_out, &a.geglu1, &a.geglu2})
          if (l->w && l->Kpad % 64 == 0 && l->Kpad == l->K && (size_t)l->N * l->Kpad * 2 >= tml_bytes) { tml.push_back(l); tm_b += (((size_t)l->N * l->Kpad * 2) + 255) & ~size_t(255); }
  LinW* kv = nullptr;  // k_proj | v_proj rows of all blocks (adjacent in the blob: the fused context projection of g_unet_forward)
  if (tml_mib > 0 && m->unet.kproj_all.w && m->unet.vproj_all.w == m->unet.kproj_all.w + (int64_t)m->unet.kproj_all.N * m->unet.kproj_all.Kpad &&
      m->unet.kproj_all.Kpad % 64 == 0 && m->unet.vproj_all.N == m->unet.kproj_all.N) {
    kv = &m->unet.kproj_all;
    tm_b += (((size_t)2 * kv->N * kv->Kpad * 2) + 255) & ~size_t(255);
  }
  // GEGLU's second linear folded into the output 1x1 convolution (AttnW::fold_w): blocks that do not run the fused tail kernel
  std::vector<AttnW*> fold;
  size_t fold_b = 0;
  if (ctx->opt.fold_out)
    for (auto& a : m->unet.attn)
      if (a.C && !attn_tail_weights_ok(a) && a.geglu2.w && a.conv_out.w && a.conv_out.k == 1 && a.conv_out.Ipad == a.C && a.conv_out.Opad == a.C &&
          a.geglu2.N == a.C && a.geglu2.Kpad == 4 * a.C && a.geglu2.K == 4 * a.C && a.C % 64 == 0) {
        fold.push_back(&a);
        fold_b += 2 * ((((size_t)a.C * 5 * a.C * 2) + 255) & ~size_t(255)) + ((((size_t)a.C * 4) + 255) & ~size_t(255));
      }
  if (el.empty() && !cin_ok && tm.empty() && tml.empty() && !kv && fold.empty()) return TSD_OK;
  const size_t tail_b = (attn_tail_stream_bytes() + 255) & ~size_t(255), head_b = (attn_head_stream_bytes() + 255) & ~size_t(255);
  const size_t each = tail_b + head_b, need = each * el.size() + cin_b + tm_b + fold_b;
  HIP_TRY(hipSetDevice(ctx->device));
  if (m->derived_bytes < need) {
    if (m->derived) HIP_TRY(hipFree(m->derived));
    m->derived = nullptr; m->derived_bytes = 0;
    hipError_t e = hipMalloc((void**)&m->derived, need);
    if (e != hipSuccess) TSD_FAIL(TSD_E_ALLOC, "derived weights: hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
    m->derived_bytes = need;
    if (ctx->opt.debug_poison >= 0 && (ctx->opt.debug_poison_what & 2)) HIP_TRY(hipMemsetAsync(m->derived, ctx->opt.debug_poison & 255, need, ctx->stream));
  }
  const bool was_planning = ctx->arena.planning;
  ctx->arena.planning = false;
  int r = TSD_OK;
  for (size_t i = 0; i < el.size() && r == TSD_OK; i++) {
    AttnW& a = *el[i];
    half_t* dst = (half_t*)(m->derived + i * each);
    if (attn_tail_weights_ok(a)) {
      r = launch_attn_tail_pack(ctx, a.sa_out.w, a.sa_out.Kpad, a.ca_q.w, a.ca_q.Kpad, a.ca_out.w, a.ca_out.Kpad, a.geglu1.w,
                                a.geglu1.Kpad, a.geglu2.w, a.geglu2.Kpad, a.conv_out.w, a.conv_out.Ipad, dst);
      if (r == TSD_OK) a.tail_stream = dst;
    }
    if (r == TSD_OK && attn_head_weights_ok(a)) {
      half_t* hd = (half_t*)(m->derived + i * each + tail_b);
      r = launch_attn_head_pack(ctx, a.conv_in.w, a.conv_in.Ipad, a.sa_in.w, a.sa_in.Kpad, hd);
      if (r == TSD_OK) a.head_stream = hd;
    }
  }
  if (r == TSD_OK && cin_ok) {
    half_t* dst = (half_t*)(m->derived + each * el.size());
    r = launch_pack_im2col_w(ctx, cin.w, cin.Opad, cin.Ipad, cin.I, dst);
    if (r == TSD_OK) m->unet.conv_in_im2col = dst;
  }
  {
    size_t off = each * el.size() + cin_b;
    for (size_t i = 0; i < tm.size() && r == TSD_OK; i++) {
      ConvW* c = tm[i];
      half_t* dst = (half_t*)(m->derived + off);
      r = launch_pack_tile_major(ctx, c->w, c->Opad, c->k * c->k * c->Ipad, dst);
      if (r == TSD_OK) c->w_tm = dst;
      off += (((size_t)c->Opad * c->k * c->k * c->Ipad * 2) + 255) & ~size_t(255);
    }
    if (kv && r == TSD_OK) {
      half_t* dst = (half_t*)(m->derived + off);
      r = launch_pack_tile_major(ctx, kv->w, 2 * kv->N, kv->Kpad, dst);
      if (r == TSD_OK) kv->w_tm = dst;
      off += (((size_t)2 * kv->N * kv->Kpad * 2) + 255) & ~size_t(255);
    }
    for (size_t i = 0; i < tml.size() && r == TSD_OK; i++) {
      LinW* l = tml[i];
      half_t* dst = (half_t*)(m->derived + off);
      r = launch_pack_tile_major(ctx, l->w, l->N, l->Kpad, dst);
      if (r == TSD_OK) l->w_tm = dst;
      off += (((size_t)l->N * l->Kpad * 2) + 255) & ~size_t(255);
    }
    for (size_t i = 0; i < fold.size() && r == TSD_OK; i++) {
      AttnW* a = fold[i];
      const int C = a->C;
      const size_t wb = (((size_t)C * 5 * C * 2) + 255) & ~size_t(255);
      half_t* wf = (half_t*)(m->derived + off);
      half_t* wf_tm = (half_t*)(m->derived + off + wb);
      float* bf = (float*)(m->derived + off + 2 * wb);
      r = launch_fold_linear_conv1x1(ctx, a->conv_out.w, a->conv_out.Ipad, a->conv_out.b, a->geglu2.w, a->geglu2.Kpad, a->geglu2.b, C, 4 * C, wf, 5 * C, bf);
      if (r == TSD_OK) r = launch_pack_tile_major(ctx, wf, C, 5 * C, wf_tm);
      if (r == TSD_OK) { a->fold_w = wf; a->fold_w_tm = wf_tm; a->fold_b = bf; }
      off += 2 * wb + ((((size_t)C * 4) + 255) & ~size_t(255));
    }
  }
  ctx->arena.planning = was_planning;
  return r;
}

int model_check_ready(tsd_model* m) {
  if (m->ready) return TSD_OK;
  for (size_t i = 0; i < m->params.size(); i++)
    if (m->params[i].used && !m->loaded[i])
      TSD_FAIL(TSD_E_STATE, "model parameter %s was never set", m->params[i].name.c_str());
  TSD_TRY(model_build_derived(m));
  m->ready = true;
  return TSD_OK;
}

extern "C" int tsd_model_prepare(tsd_model* m) {
  if (!m) TSD_FAIL(TSD_E_ARG, "NULL model");
  HIP_TRY(hipSetDevice(m->ctx->device));
  TSD_TRY(model_check_ready(m));
  HIP_TRY(hipStreamSynchronize(m->ctx->stream));
  return TSD_OK;
}

ConvW model_conv(const tsd_model* m, const std::string& prefix) {
  ConvW w;
  auto it = m->index.find(prefix + ".kernel");
  if (it == m->index.end()) return w;
  const ParamSpec& pw = m->params[it->second];
  const ParamSpec& pb = m->params[it->second + 1];
  if (!pw.used) return w;
  w.w = (const half_t*)(m->blob + pw.off);
  w.b = (const float*)(m->blob + pb.off);
  w.O = (int)pw.shape[0]; w.I = (int)pw.shape[1]; w.k = (int)pw.shape[2]; w.Ipad = pw.Kpad; w.Opad = pw.Opad;
  return w;
}

LinW model_lin(const tsd_model* m, const std::string& prefix, bool use_bias) {
  LinW w;
  auto it = m->index.find(prefix + ".weight");
  if (it == m->index.end()) return w;
  const ParamSpec& pw = m->params[it->second];
  const ParamSpec& pb = m->params[it->second + 1];
  w.w = (const half_t*)(m->blob + pw.off);
  w.b = (use_bias && pb.used) ? (const float*)(m->blob + pb.off) : nullptr;
  w.N = (int)pw.shape[0]; w.K = (int)pw.shape[1]; w.Kpad = pw.Kpad;
  return w;
}

int model_resolve(tsd_model* m) {
  if (is_diffusion_kind(m->kind)) {
    UNetW& u = m->unet;
    const bool full = is_full_unet_kind(m->kind);
    const bool torch_norms = m->kind == TSD_MODEL_DIFFUSION_SD15_TORCH;
    auto aff = [&](const std::string& name) {
      NormAffine a;
      if (torch_norms) {
        a.w = (const float*)(m->blob + m->params[m->index.at(name + ".weight")].off);
        a.b = (const float*)(m->blob + m->params[m->index.at(name + ".bias")].off);
        a.torch_rstd = 1;
      }
      return a;
    };
    const int n_layers = full ? SD15_N : 23;
    u.res.assign(n_layers, ResW());
    u.attn.assign(n_layers, AttnW());
    u.conv.assign(n_layers, ConvW());
    u.t1 = model_lin(m, "time_embed.layer1", true);
    u.t2 = model_lin(m, "time_embed.layer2", true);
    int toff = 0, kvoff = 0;
    bool first = true;
    for (int i = 0; i < n_layers; i++) {
      const LayerDef& l = full ? SD15_STEPS[i].l : UNET_LAYERS[i];
      const std::string n = "unet.layer" + std::to_string(i + 1);
      if (l.kind == L_RES) {
        ResW& r = u.res[i];
        r.cin = l.a; r.cout = l.b; r.groups = 32; r.has_skip = l.a != l.b;
        r.conv1 = model_conv(m, n + ".layer2");
        r.conv2 = model_conv(m, n + ".layer5");
        if (r.has_skip) r.skip = model_conv(m, n + ".layer6");
        r.time = model_lin(m, n + ".layer3", true);
        r.time_off = toff;
        if (first) { u.tproj = r.time; first = false; }
        toff += l.b;
        r.gn1 = aff(n + ".layer1"); r.gn2 = aff(n + ".layer4");
      } else if (l.kind == L_ATTN) {
        AttnW& a = u.attn[i];
        a.n_head = l.a; a.n_embed = l.b; a.C = l.a * l.b;
        a.conv_in = model_conv(m, n + ".layer2");
        a.sa_in = model_lin(m, n + ".layer4.in_proj", false);
        a.sa_out = model_lin(m, n + ".layer4.out_proj", true);
        a.ca_q = model_lin(m, n + ".layer6.q_proj", false);
        a.ca_k = model_lin(m, n + ".layer6.k_proj", false);
        a.ca_v = model_lin(m, n + ".layer6.v_proj", false);
        a.ca_out = model_lin(m, n + ".layer6.out_proj", true);
        a.geglu1 = model_lin(m, n + ".layer8", true);
        a.geglu2 = model_lin(m, n + ".layer9", true);
        a.conv_out = model_conv(m, n + ".layer10");
        a.kv_off = kvoff;
        if (kvoff == 0) { u.kproj_all = a.ca_k; u.vproj_all = a.ca_v; }
        kvoff += a.C;
        a.gelu_erf = torch_norms;
        a.gn = aff(n + ".layer1"); a.ln[0] = aff(n + ".layer3"); a.ln[1] = aff(n + ".layer5"); a.ln[2] = aff(n + ".layer7");
      } else if (full && (l.kind == L_CONV || l.kind == L_UPCONV)) {
        u.conv[i] = model_conv(m, n);
      }
    }
    u.tproj.N = toff;  // the layer3 weights of all residual blocks are contiguous in the blob (6720 rows; 20160 full-size)
    u.kproj_all.N = kvoff; u.vproj_all.N = kvoff;  // 6720 rows each (12480 full-size)
    if (!full) {
      u.conv1 = model_conv(m, "unet.layer1");
      u.conv4 = model_conv(m, "unet.layer4");
      u.conv7 = model_conv(m, "unet.layer7");
    }
    u.final_conv = model_conv(m, "final.layer2");
    u.final_gn = aff("final.layer1");
    u.final_groups = torch_norms ? 32 : 320;
  } else if (is_clip_kind(m->kind)) {
    ClipW& c = m->clip;
    auto aff = [&](const std::string& name) {
      NormAffine a;
      if (m->kind == TSD_MODEL_CLIP_TORCH) {
        a.w = (const float*)(m->blob + m->params[m->index.at(name + ".weight")].off);
        a.b = (const float*)(m->blob + m->params[m->index.at(name + ".bias")].off);
        a.torch_rstd = 1;
      }
      return a;
    };
    c.final_ln = aff("layernorm");
    c.tok = (const half_t*)(m->blob + m->params[m->index.at("embedding.token.weight")].off);
    c.pos = (const float*)(m->blob + m->params[m->index.at("embedding.position")].off);
    for (int i = 0; i < 12; i++) {
      const std::string n = "player" + std::to_string(i + 1);
      c.layer[i].in_proj = model_lin(m, n + ".layer2.in_proj", true);
      c.layer[i].out_proj = model_lin(m, n + ".layer2.out_proj", true);
      c.layer[i].l4 = model_lin(m, n + ".layer4", true);
      c.layer[i].l5 = model_lin(m, n + ".layer5", true);
      c.layer[i].ln1 = aff(n + ".layer1");
      c.layer[i].ln2 = aff(n + ".layer3");
    }
  } else {
    const LayerDef* L = is_decoder_kind(m->kind) ? DECODER_LAYERS : ENCODER_LAYERS;
    const int n_layers = is_decoder_kind(m->kind) ? 26 : 19;
    const bool torch_norms = is_vae_torch_kind(m->kind);
    auto aff = [&](const std::string& name) {
      NormAffine a;
      if (torch_norms) {
        a.w = (const float*)(m->blob + m->params[m->index.at(name + ".weight")].off);
        a.b = (const float*)(m->blob + m->params[m->index.at(name + ".bias")].off);
        a.torch_rstd = 1;
      }
      return a;
    };
    VaeW& v = m->vae;
    v.gn_eps = torch_norms ? 1e-6f : 1e-5f;
    v.gn.assign(n_layers, NormAffine());
    v.conv.assign(n_layers, ConvW());
    v.res.assign(n_layers, ResW());
    v.attn.assign(n_layers, VaeAttnW());
    for (int i = 0; i < n_layers; i++) {
      const LayerDef& l = L[i];
      const std::string n = "l" + std::to_string(i + 1);
      if (l.kind == L_CONV || l.kind == L_CONV_S2) v.conv[i] = model_conv(m, n);
      else if (l.kind == L_RES) {
        ResW& r = v.res[i];
        r.cin = l.a; r.cout = l.b; r.has_skip = l.a != l.b;
        r.groups = torch_norms ? 32 : 16;  // GroupNorm(16), vae.mojo:42-43 ; the trained VAE has 32
        r.eps = torch_norms ? 1e-6f : 1e-5f;
        r.gn1 = aff(n + ".group_norm1"); r.gn2 = aff(n + ".group_norm2");
        r.conv1 = model_conv(m, n + ".conv1");
        r.conv2 = model_conv(m, n + ".conv2");
        if (r.has_skip) r.skip = model_conv(m, n + ".res_conv_layer");
      } else if (l.kind == L_ATTN) {
        v.attn[i].C = l.a;
        v.attn[i].in_proj = model_lin(m, n + ".attention.in_proj", true);
        v.attn[i].out_proj = model_lin(m, n + ".attention.out_proj", true);
        v.attn[i].gn = aff(n + ".group_norm");
        v.attn[i].eps = torch_norms ? 1e-6f : 1e-5f;
      } else if (l.kind == L_GN) {
        v.gn[i] = aff(n);
      }
    }
  }
  return TSD_OK;
}
