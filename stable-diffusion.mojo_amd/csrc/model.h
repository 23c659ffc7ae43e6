// model.h - parameter inventory, packed device weights and resolved weight trees.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "common.h"

enum ParamKind { P_CONV_W = 0, P_CONV_B = 1, P_LIN_W = 2, P_LIN_B = 3 };

struct ParamSpec {
  std::string name;
  int64_t shape[4] = {0, 0, 0, 0};
  int ndim = 0;
  int kind = 0;
  bool used = true;
  float bound = 0.f;  // synthetic-init bound (0 -> zeros)
  float offset = 0.f; // synthetic init = offset + U(+-bound) (norm weights: 1 + ...)
  // packing
  int Opad = 0, Kpad = 0;  // conv: Opad rows, Kpad = Ipad ; linear: Kpad
  int interleave = 0;      // GEGLU (a,g) row interleave (diffusion.mojo:138-141)
  int region = 0;          // 0 normal; dense contiguous tables: 1 time-proj weights, 2 time-proj biases, 3 k_proj, 4 v_proj
  size_t off = 0, bytes = 0;
  int64_t numel() const {
    int64_t n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    return n;
  }
};

// layer tables (diffusion.mojo:177-201, vae.mojo:94-112,194-219)
enum LayerKind { L_CONV, L_CONV_S2, L_RES, L_ATTN, L_UP, L_GN, L_SILU, L_UPCONV };
struct LayerDef { int kind; int a, b, c, d; };
extern const LayerDef UNET_LAYERS[23];
extern const LayerDef DECODER_LAYERS[26];
extern const LayerDef ENCODER_LAYERS[19];
// full-size UNet (TSD_MODEL_DIFFUSION_SD15): flat layer list with the skip-connection plumbing of each step
enum { U_POP = 1, U_PUSH = 2 };  // input = concat(x, popped skip) ; output pushed as a skip
struct UNetStep { LayerDef l; int flags; };
constexpr int SD15_N = 45;
extern const UNetStep SD15_STEPS[SD15_N];
static inline bool is_full_unet_kind(int kind) { return kind == TSD_MODEL_DIFFUSION_SD15 || kind == TSD_MODEL_DIFFUSION_SD15_TORCH; }
static inline bool is_diffusion_kind(int kind) { return kind == TSD_MODEL_DIFFUSION || is_full_unet_kind(kind); }
static inline bool is_clip_kind(int kind) { return kind == TSD_MODEL_CLIP || kind == TSD_MODEL_CLIP_TORCH; }
static inline bool is_decoder_kind(int kind) { return kind == TSD_MODEL_DECODER || kind == TSD_MODEL_DECODER_TORCH; }
static inline bool is_encoder_kind(int kind) { return kind == TSD_MODEL_ENCODER || kind == TSD_MODEL_ENCODER_TORCH; }
static inline bool is_vae_torch_kind(int kind) { return kind == TSD_MODEL_DECODER_TORCH || kind == TSD_MODEL_ENCODER_TORCH; }
constexpr int TSD_MODEL_KIND_MAX = TSD_MODEL_ENCODER_TORCH;

std::vector<ParamSpec> build_param_specs(int model_kind);

struct ResW {
  ConvW conv1, conv2, skip;
  LinW time;       // UNet only (kept for the op-level path; the model path uses the concatenated table)
  int cin = 0, cout = 0, groups = 32;
  bool has_skip = false;
  int time_off = 0;  // column offset into the concatenated time projection [B][6720]
  NormAffine gn1, gn2;  // torch-norm extension (kind 6): per-channel affine of the two GroupNorms (w == nullptr: reference)
  float eps = 1e-5f;    // GroupNorm epsilon (helpers/utils.mojo:1821-1826 default); 1e-6 in a trained VAE (kinds 8, 9)
};
struct AttnW {  // Unet_Attention_Block, diffusion.mojo:87-98
  int n_head = 0, n_embed = 0, C = 0, d_ctx = 768;
  ConvW conv_in, conv_out;
  LinW sa_in, sa_out, ca_q, ca_k, ca_v, ca_out, geglu1, geglu2;
  int kv_off = 0;  // row offset of this block in the concatenated k_proj / v_proj tables
  NormAffine gn, ln[3];  // torch-norm extension (kind 6)
  bool gelu_erf = false; // kind 6: exact GELU in the GEGLU gate (torch.nn.functional.gelu)
  // derived: the six tail matrices re-packed as the tile stream of the fused tail kernel (kernels_chain.hip); nullptr when
  // the block is not eligible or the model's parameters changed since the last model_check_ready()
  const half_t* tail_stream = nullptr;
  const half_t* head_stream = nullptr;  // same for conv_in + in_proj (fused head kernel)
  // derived, blocks that run op by op (C = 640 / 1280): `conv_out(geglu2(h) + r) + x` (diffusion.mojo:143-146) is linear in [h | r], so the two
  // GEMMs run as ONE over the channel concat with folded weights [C][4C + C] (row-major and K-tile-major) and bias conv_out.w . geglu2.b + conv_out.b
  const half_t* fold_w = nullptr; const half_t* fold_w_tm = nullptr; const float* fold_b = nullptr;
};
// true when the fused tail kernel can run this block's weights (reference norms, tanh GELU, C = 8 x 40)
bool attn_tail_weights_ok(const AttnW& w);
bool attn_head_weights_ok(const AttnW& w);
struct VaeAttnW {  // vae.mojo:9-11
  int C = 0;
  LinW in_proj, out_proj;
  NormAffine gn;  // torch-norm extension (kinds 8, 9)
  float eps = 1e-5f;
};

struct UNetW {
  LinW t1, t2;       // Time_Embedding
  LinW tproj;        // concatenated layer3 of the 9 residual blocks: [6720][1280]
  LinW kproj_all, vproj_all;  // concatenated cross-attention k_proj / v_proj of the 9 attention blocks: [6720][768]
  ConvW conv1, conv4, conv7, final_conv;
  // derived: the input convolution's weights as a [Opad][64] matrix over K = (tap, cin) for cin < 8 (model_check_ready): the 4-channel
  // latent is gathered to im2col rows at the boundary and the convolution runs as ONE 64-deep K tile instead of nine padded taps
  const half_t* conv_in_im2col = nullptr;
  std::vector<ResW> res;    // indexed by flat layer position (23 entries, or SD15_N)
  std::vector<AttnW> attn;
  std::vector<ConvW> conv;  // full-size UNet only: input, downsample and upsample convolutions
  NormAffine final_gn;      // torch-norm extension (kind 6)
  int final_groups = 320;   // UNet_Output_Layer's GroupNorm (diffusion.mojo:287-291); 32 with torch norms
};
struct VaeW {
  std::vector<ConvW> conv;      // indexed by layer (1-based position - 1)
  const half_t* conv_in_im2col = nullptr;  // derived: the first convolution (3 or 4 input channels) as [Opad][64] im2col weights
  std::vector<ResW> res;
  std::vector<VaeAttnW> attn;
  std::vector<NormAffine> gn;   // stand-alone GroupNorm layers (torch-norm extension)
  float gn_eps = 1e-5f;         // their epsilon (1e-6 in a trained VAE)
};

struct ClipLayerW { LinW in_proj, out_proj, l4, l5; NormAffine ln1, ln2; };  // ClipPlayer clip.mojo:23-34 (its LayerNorms have no parameters)
struct ClipW {
  const half_t* tok = nullptr;  // [49408][768] fp16
  const float* pos = nullptr;   // [77][768] fp32
  ClipLayerW layer[12];
  NormAffine final_ln;  // torch-norm extension (kind 7)
};

struct PlanKey { int B, L, T; bool operator<(const PlanKey& o) const { return B != o.B ? B < o.B : (L != o.L ? L < o.L : T < o.T); } };

struct tsd_model {
  tsd_ctx* ctx = nullptr;
  int kind = 0;
  std::vector<ParamSpec> params;
  std::map<std::string, int> index;
  char* blob = nullptr;
  size_t blob_bytes = 0;
  std::vector<char> loaded;
  bool ready = false;
  UNetW unet;
  VaeW vae;
  ClipW clip;
  std::map<PlanKey, size_t> plans;  // workspace high-water mark per problem shape
  char* derived = nullptr;          // derived device buffers rebuilt by model_check_ready(): fused-tail weight streams
  size_t derived_bytes = 0;
};

int model_resolve(tsd_model* m);                      // fill unet / vae from the packed blob
ConvW model_conv(const tsd_model* m, const std::string& prefix);
LinW model_lin(const tsd_model* m, const std::string& prefix, bool use_bias);
int model_check_ready(tsd_model* m);
