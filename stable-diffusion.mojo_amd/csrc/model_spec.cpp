// model_spec.cpp - parameter inventory of the reference models in struct-field DFS order
// (`Diffusion` diffusion.mojo:299-302 / :175-201 ; `Decoder` vae.mojo:194-219 ; `Encoder`
// vae.mojo:94-112), the packing decisions per parameter, and the algorithmic FLOP census.
#include <math.h>
#include <string.h>

#include "model.h"

// {kind, a, b, c, d}: conv (cin,cout,k,stride) ; res (cin,cout) ; attn (n_head,n_embed) / (C)
const LayerDef UNET_LAYERS[23] = {
    {L_CONV, 4, 320, 3, 1},   {L_RES, 320, 320},   {L_ATTN, 8, 40},  {L_CONV, 320, 320, 3, 2}, {L_RES, 320, 640},
    {L_ATTN, 8, 80},          {L_CONV, 640, 640, 3, 2}, {L_RES, 640, 1280}, {L_ATTN, 8, 160},  {L_RES, 2560, 1280},
    {L_ATTN, 8, 160},         {L_RES, 1920, 1280}, {L_ATTN, 8, 160}, {L_UP},                   {L_RES, 1280, 640},
    {L_ATTN, 8, 80},          {L_RES, 960, 640},   {L_ATTN, 8, 80},  {L_UP},                   {L_RES, 640, 320},
    {L_ATTN, 8, 40},          {L_RES, 640, 320},   {L_ATTN, 8, 40}};

// Full-size UNet: encoders push their output, every decoder block starts from concat(x, popped skip).
#define R(ci, co, f) {{L_RES, ci, co, 0, 0}, f}
#define A(dh, f) {{L_ATTN, 8, dh, 0, 0}, f}
const UNetStep SD15_STEPS[SD15_N] = {
    {{L_CONV, 4, 320, 3, 1}, U_PUSH},
    R(320, 320, 0), A(40, U_PUSH), R(320, 320, 0), A(40, U_PUSH), {{L_CONV, 320, 320, 3, 2}, U_PUSH},
    R(320, 640, 0), A(80, U_PUSH), R(640, 640, 0), A(80, U_PUSH), {{L_CONV, 640, 640, 3, 2}, U_PUSH},
    R(640, 1280, 0), A(160, U_PUSH), R(1280, 1280, 0), A(160, U_PUSH), {{L_CONV, 1280, 1280, 3, 2}, U_PUSH},
    R(1280, 1280, U_PUSH), R(1280, 1280, U_PUSH),
    R(1280, 1280, 0), A(160, 0), R(1280, 1280, 0),  // bottleneck
    R(2560, 1280, U_POP), R(2560, 1280, U_POP), R(2560, 1280, U_POP), {{L_UPCONV, 1280, 1280, 3, 1}, 0},
    R(2560, 1280, U_POP), A(160, 0), R(2560, 1280, U_POP), A(160, 0), R(1920, 1280, U_POP), A(160, 0),
    {{L_UPCONV, 1280, 1280, 3, 1}, 0},
    R(1920, 640, U_POP), A(80, 0), R(1280, 640, U_POP), A(80, 0), R(960, 640, U_POP), A(80, 0),
    {{L_UPCONV, 640, 640, 3, 1}, 0},
    R(960, 320, U_POP), A(40, 0), R(640, 320, U_POP), A(40, 0), R(640, 320, U_POP), A(40, 0)};
#undef R
#undef A

const LayerDef DECODER_LAYERS[26] = {
    {L_CONV, 4, 4, 1, 1},     {L_CONV, 4, 512, 3, 1},   {L_RES, 512, 512}, {L_ATTN, 512},     {L_RES, 512, 512},
    {L_RES, 512, 512},        {L_RES, 512, 512},        {L_RES, 512, 512}, {L_UP},            {L_CONV, 512, 512, 3, 1},
    {L_RES, 512, 512},        {L_RES, 512, 512},        {L_RES, 512, 512}, {L_UP},            {L_CONV, 512, 512, 3, 1},
    {L_RES, 512, 256},        {L_RES, 256, 256},        {L_RES, 256, 256}, {L_UP},            {L_CONV, 256, 256, 3, 1},
    {L_RES, 256, 128},        {L_RES, 128, 128},        {L_RES, 128, 128}, {L_GN, 32, 128},   {L_SILU},
    {L_CONV, 128, 3, 3, 1}};

const LayerDef ENCODER_LAYERS[19] = {
    {L_CONV, 3, 128, 3, 1},   {L_RES, 128, 128}, {L_RES, 128, 128}, {L_CONV_S2, 128, 128, 3, 2}, {L_RES, 128, 256},
    {L_RES, 256, 256},        {L_CONV_S2, 256, 256, 3, 2}, {L_RES, 256, 512}, {L_RES, 512, 512},
    {L_CONV_S2, 512, 512, 3, 2}, {L_RES, 512, 512}, {L_RES, 512, 512}, {L_RES, 512, 512}, {L_ATTN, 512},
    {L_RES, 512, 512},        {L_GN, 32, 512},   {L_SILU},          {L_CONV, 512, 8, 3, 1},      {L_CONV, 8, 8, 1, 1}};

namespace {
struct Builder {
  std::vector<ParamSpec> out;
  void conv(const std::string& name, int cin, int cout, int k, bool used = true, bool last = false) {
    ParamSpec w;
    w.name = name + ".kernel"; w.ndim = 4; w.shape[0] = cout; w.shape[1] = cin; w.shape[2] = k; w.shape[3] = k;
    w.kind = P_CONV_W; w.used = used;
    w.bound = (float)(1.0 / sqrt((double)cin * k * k));  // helpers/utils.mojo:1722-1724
    // an output narrower than one K chunk that feeds another conv is padded with exact zeros
    w.Opad = (cout < 64 && !last) ? 64 : round_up(cout, 4);
    w.Kpad = round_up(cin, 64);
    out.push_back(w);
    ParamSpec b;
    b.name = name + ".bias"; b.ndim = 1; b.shape[0] = cout; b.kind = P_CONV_B; b.used = used;
    b.bound = 0.f;  // Tensor zero-init, helpers/utils.mojo:1717 (App.A D17)
    b.Opad = w.Opad;
    out.push_back(b);
  }
  void lin(const std::string& name, int fin, int fout, bool use_bias = true, bool used = true, int interleave = 0,
           int region = 0) {
    ParamSpec w;
    w.name = name + ".weight"; w.ndim = 2; w.shape[0] = fout; w.shape[1] = fin; w.kind = P_LIN_W; w.used = used;
    w.bound = (float)(1.0 / sqrt((double)fin));  // App.A D18 (1/sqrt(fan_in), not the literal in^-1/4)
    w.Kpad = round_up(fin, 64); w.Opad = fout; w.interleave = interleave; w.region = region;
    out.push_back(w);
    ParamSpec b;
    b.name = name + ".bias"; b.ndim = 1; b.shape[0] = fout; b.kind = P_LIN_B; b.used = used && use_bias;
    b.bound = w.bound; b.Opad = fout; b.interleave = interleave; b.region = region == 1 ? 2 : 0;
    out.push_back(b);
  }
  void norm(const std::string& name, int C) {  // torch-norm extension: per-channel weight (1 + U) and bias
    ParamSpec w;
    w.name = name + ".weight"; w.ndim = 1; w.shape[0] = C; w.kind = P_LIN_B; w.bound = 0.3f; w.offset = 1.f; w.Opad = C;
    out.push_back(w);
    ParamSpec b;
    b.name = name + ".bias"; b.ndim = 1; b.shape[0] = C; b.kind = P_LIN_B; b.bound = 0.2f; b.Opad = C;
    out.push_back(b);
  }
  void unet_res(const std::string& n, int cin, int cout) {  // diffusion.mojo:34-42
    conv(n + ".layer2", cin, cout, 3);
    lin(n + ".layer3", 1280, cout, true, true, 0, 1);
    conv(n + ".layer5", cout, cout, 3);
    conv(n + ".layer6", cin, cout, 1, cin != cout);
  }
  void unet_attn(const std::string& n, int nh, int ne, int dctx = 768) {  // diffusion.mojo:87-98
    const int C = nh * ne;
    conv(n + ".layer2", C, C, 1);
    lin(n + ".layer4.in_proj", C, 3 * C, false);
    lin(n + ".layer4.out_proj", C, C);
    lin(n + ".layer6.q_proj", C, C, false);
    lin(n + ".layer6.k_proj", dctx, C, false, true, 0, 3);
    lin(n + ".layer6.v_proj", dctx, C, false, true, 0, 4);
    lin(n + ".layer6.out_proj", C, C);
    lin(n + ".layer8", C, 8 * C, true, true, 1);
    lin(n + ".layer9", 4 * C, C);
    conv(n + ".layer10", C, C, 1);
  }
  void vae_res(const std::string& n, int cin, int cout) {  // vae.mojo:39-46
    conv(n + ".conv1", cin, cout, 3);
    conv(n + ".conv2", cout, cout, 3);
    conv(n + ".res_conv_layer", cin, cout, 1, cin != cout);
  }
  void vae_attn(const std::string& n, int C) {  // vae.mojo:9-11
    lin(n + ".attention.in_proj", C, 3 * C);
    lin(n + ".attention.out_proj", C, C);
  }
};
}  // namespace

std::vector<ParamSpec> build_param_specs(int kind) {
  Builder b;
  if (kind == TSD_MODEL_DIFFUSION) {
    b.lin("time_embed.layer1", 320, 1280);
    b.lin("time_embed.layer2", 1280, 1280);
    for (int i = 0; i < 23; i++) {
      const LayerDef& l = UNET_LAYERS[i];
      const std::string n = "unet.layer" + std::to_string(i + 1);
      if (l.kind == L_CONV) b.conv(n, l.a, l.b, l.c);
      else if (l.kind == L_RES) b.unet_res(n, l.a, l.b);
      else if (l.kind == L_ATTN) b.unet_attn(n, l.a, l.b);
    }
    b.conv("final.layer2", 320, 4, 3, true, true);
  } else if (is_full_unet_kind(kind)) {  // same blocks, full layer list; names "unet.layerN" by flat position
    b.lin("time_embed.layer1", 320, 1280);
    b.lin("time_embed.layer2", 1280, 1280);
    for (int i = 0; i < SD15_N; i++) {
      const LayerDef& l = SD15_STEPS[i].l;
      const std::string n = "unet.layer" + std::to_string(i + 1);
      if (l.kind == L_CONV || l.kind == L_UPCONV) b.conv(n, l.a, l.b, l.c);
      else if (l.kind == L_RES) b.unet_res(n, l.a, l.b);
      else if (l.kind == L_ATTN) b.unet_attn(n, l.a, l.b);
    }
    b.conv("final.layer2", 320, 4, 3, true, true);
    if (kind == TSD_MODEL_DIFFUSION_SD15_TORCH) {  // norm parameters appended, so the shared indices equal kind 5's
      for (int i = 0; i < SD15_N; i++) {
        const LayerDef& l = SD15_STEPS[i].l;
        const std::string n = "unet.layer" + std::to_string(i + 1);
        if (l.kind == L_RES) { b.norm(n + ".layer1", l.a); b.norm(n + ".layer4", l.b); }
        else if (l.kind == L_ATTN) {
          const int C = l.a * l.b;
          b.norm(n + ".layer1", C); b.norm(n + ".layer3", C); b.norm(n + ".layer5", C); b.norm(n + ".layer7", C);
        }
      }
      b.norm("final.layer1", 320);
    }
  } else if (is_clip_kind(kind)) {  // clip.mojo:74-88 ; parameter order = oracle/spec.py clip_params()
    ParamSpec t;
    t.name = "embedding.token.weight"; t.ndim = 2; t.shape[0] = 49408; t.shape[1] = 768; t.kind = P_LIN_W;
    t.bound = 1.7320508075688772f;  // unit variance like init_weights_normal(0,1), helpers/utils.mojo:2025
    t.Kpad = 768; t.Opad = 49408;
    b.out.push_back(t);
    ParamSpec p;
    p.name = "embedding.position"; p.ndim = 1; p.shape[0] = 77 * 768; p.kind = P_LIN_B; p.bound = 0.02f; p.Opad = 77 * 768;
    b.out.push_back(p);
    for (int i = 1; i <= 12; i++) {
      const std::string n = "player" + std::to_string(i);
      b.lin(n + ".layer2.in_proj", 768, 3 * 768);
      b.lin(n + ".layer2.out_proj", 768, 768);
      b.lin(n + ".layer4", 768, 4 * 768);
      b.lin(n + ".layer5", 4 * 768, 768);
    }
    if (kind == TSD_MODEL_CLIP_TORCH) {
      for (int i = 1; i <= 12; i++) {
        b.norm("player" + std::to_string(i) + ".layer1", 768);
        b.norm("player" + std::to_string(i) + ".layer3", 768);
      }
      b.norm("layernorm", 768);
    }
  } else if (is_decoder_kind(kind) || is_encoder_kind(kind)) {
    const LayerDef* L = is_decoder_kind(kind) ? DECODER_LAYERS : ENCODER_LAYERS;
    const int n_layers = is_decoder_kind(kind) ? 26 : 19;
    for (int i = 0; i < n_layers; i++) {
      const LayerDef& l = L[i];
      const std::string n = "l" + std::to_string(i + 1);
      if (l.kind == L_CONV || l.kind == L_CONV_S2) b.conv(n, l.a, l.b, l.c, true, i == n_layers - 1);
      else if (l.kind == L_RES) b.vae_res(n, l.a, l.b);
      else if (l.kind == L_ATTN) b.vae_attn(n, l.a);
    }
    if (is_vae_torch_kind(kind)) {  // norm parameters appended, so the shared indices equal the reference kind's
      for (int i = 0; i < n_layers; i++) {
        const LayerDef& l = L[i];
        const std::string n = "l" + std::to_string(i + 1);
        if (l.kind == L_RES) { b.norm(n + ".group_norm1", l.a); b.norm(n + ".group_norm2", l.b); }
        else if (l.kind == L_ATTN) b.norm(n + ".group_norm", l.a);
        else if (l.kind == L_GN) b.norm(n, l.b);
      }
    }
  }
  return b.out;
}

extern "C" int tsd_model_param_count(int kind) {
  if (kind < TSD_MODEL_DIFFUSION || kind > TSD_MODEL_KIND_MAX) return TSD_E_ARG;
  return (int)build_param_specs(kind).size();
}

extern "C" int tsd_model_param_info(int kind, int index, char* name, int name_cap, int64_t shape[4], int* ndim,
                                    int* used, float* init_bound) {
  if (kind < TSD_MODEL_DIFFUSION || kind > TSD_MODEL_KIND_MAX) TSD_FAIL(TSD_E_ARG, "bad model kind %d", kind);
  static thread_local int cached_kind = 0;
  static thread_local std::vector<ParamSpec> cached;
  if (cached_kind != kind) {
    cached = build_param_specs(kind);
    cached_kind = kind;
  }
  if (index < 0 || index >= (int)cached.size()) TSD_FAIL(TSD_E_ARG, "param index %d out of range", index);
  const ParamSpec& p = cached[index];
  if (name && name_cap > 0) {
    strncpy(name, p.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (shape) for (int i = 0; i < 4; i++) shape[i] = i < p.ndim ? p.shape[i] : 1;
  if (ndim) *ndim = p.ndim;
  if (used) *used = p.used ? 1 : 0;
  if (init_bound) *init_bound = p.bound;
  return TSD_OK;
}

// ---- algorithmic FLOP census (SURVEY.md Appendix B method): 2*MAC of conv + linear + attention core
namespace {
double conv_f(double cin, double cout, double k, double hw) { return 2.0 * cin * cout * k * k * hw; }
double lin_f(double m, double k, double n) { return 2.0 * m * k * n; }
double attn_core_f(double H, double tq, double tk, double dh) { return 2.0 * 2.0 * H * tq * tk * dh; }
double unet_res_f(double cin, double cout, double hw) {
  return conv_f(cin, cout, 3, hw) + lin_f(1, 1280, cout) + conv_f(cout, cout, 3, hw) +
         (cin != cout ? conv_f(cin, cout, 1, hw) : 0.0);
}
double unet_attn_f(double C, double dh, double hw, double T) {
  const double H = C / dh;
  return 2 * conv_f(C, C, 1, hw) + lin_f(hw, C, 3 * C) + attn_core_f(H, hw, hw, dh) + lin_f(hw, C, C) +
         lin_f(hw, C, C) + 2 * lin_f(T, 768, C) + attn_core_f(H, hw, T, dh) + lin_f(hw, C, C) + lin_f(hw, C, 8 * C) +
         lin_f(hw, 4 * C, C);
}
double vae_res_f(double cin, double cout, double hw) {
  return conv_f(cin, cout, 3, hw) + conv_f(cout, cout, 3, hw) + (cin != cout ? conv_f(cin, cout, 1, hw) : 0.0);
}
double vae_attn_f(double C, double hw) { return lin_f(hw, C, 3 * C) + attn_core_f(1, hw, hw, C) + lin_f(hw, C, C); }
}  // namespace

extern "C" double tsd_flop_count(int kind, int L, int T) {
  double f = 0.0;
  if (kind == TSD_MODEL_DIFFUSION) {
    f += lin_f(1, 320, 1280) + lin_f(1, 1280, 1280);
    double side = L;
    for (int i = 0; i < 23; i++) {
      const LayerDef& l = UNET_LAYERS[i];
      if (l.kind == L_CONV) {
        if (l.d == 2) side /= 2;
        f += conv_f(l.a, l.b, l.c, side * side);
      } else if (l.kind == L_RES) f += unet_res_f(l.a, l.b, side * side);
      else if (l.kind == L_ATTN) f += unet_attn_f((double)l.a * l.b, l.b, side * side, T);
      else if (l.kind == L_UP) side *= 2;
    }
    f += conv_f(320, 4, 3, (double)L * L);
  } else if (is_full_unet_kind(kind)) {
    f += lin_f(1, 320, 1280) + lin_f(1, 1280, 1280);
    double side = L;
    for (int i = 0; i < SD15_N; i++) {
      const LayerDef& l = SD15_STEPS[i].l;
      if (l.kind == L_CONV) {
        if (l.d == 2) side /= 2;
        f += conv_f(l.a, l.b, l.c, side * side);
      } else if (l.kind == L_UPCONV) {
        side *= 2;
        f += conv_f(l.a, l.b, l.c, side * side);
      } else if (l.kind == L_RES) f += unet_res_f(l.a, l.b, side * side);
      else if (l.kind == L_ATTN) f += unet_attn_f((double)l.a * l.b, l.b, side * side, T);
    }
    f += conv_f(320, 4, 3, (double)L * L);
  } else if (is_decoder_kind(kind) || is_encoder_kind(kind)) {
    const LayerDef* Ls = is_decoder_kind(kind) ? DECODER_LAYERS : ENCODER_LAYERS;
    const int n = is_decoder_kind(kind) ? 26 : 19;
    double side = L;  // decoder: latent side ; encoder: image side
    for (int i = 0; i < n; i++) {
      const LayerDef& l = Ls[i];
      if (l.kind == L_CONV) f += conv_f(l.a, l.b, l.c, side * side);
      else if (l.kind == L_CONV_S2) { side /= 2; f += conv_f(l.a, l.b, l.c, side * side); }
      else if (l.kind == L_RES) f += vae_res_f(l.a, l.b, side * side);
      else if (l.kind == L_ATTN) f += vae_attn_f(l.a, side * side);
      else if (l.kind == L_UP) side *= 2;
    }
  } else if (is_clip_kind(kind)) {  // per prompt: 12 x (in_proj, QK^T, PV, out_proj, 768->3072->768) on 77 tokens
    const double Tc = 77, D = 768;
    f = 12.0 * (lin_f(Tc, D, 3 * D) + 2.0 * 2.0 * Tc * Tc * D + lin_f(Tc, D, D) + lin_f(Tc, D, 4 * D) + lin_f(Tc, 4 * D, D));
  } else {
    return -1.0;
  }
  return f / 1e9;
}
