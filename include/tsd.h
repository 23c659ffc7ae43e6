/* tsd.h - C ABI of libtsd.so, the MI355X-native Tiny-Stable-Diffusion hot path.
 *
 * This header is the drop-in boundary: plain C scalars and pointers only,
 * no structs by value, no callbacks, so the reference's host language (Mojo, via
 * sys.ffi.DLHandle / external_call) or any other FFI can bind it.  The reference has no
 * FFI of its own; each entry point below replaces one Mojo struct's `forward()` and cites it
 * (paths relative to the reference repo lrmantovani10/Stable-Diffusion.mojo).
 *
 * Conventions
 *   - Host tensors are float32 in the reference's own layouts (`Matrix._data`,
 *     helpers/utils.mojo:805-811): images/activations CHW, token tensors (T, D), batched
 *     tensors contiguous [B][...] (`Matrix_Array`, helpers/utils.mojo:464-468).
 *   - The caller owns every input and output buffer; the library never keeps a caller pointer
 *     after return and never mutates an input (the reference's accidental aliasing,
 *     SURVEY.md App.A D14/D16, is not reproduced).
 *   - Arithmetic on the device is fp16 storage / fp32 accumulate (MFMA) - the "_f32" suffix
 *     names the boundary dtype, not the compute precision.
 *   - Every function returns 0 (TSD_OK) or a negative tsd_status; `tsd_last_error()` gives
 *     the thread-local message.  The reference prints and returns a null Matrix on shape
 *     errors (e.g. helpers/utils.mojo:1955-1957); a shim maps non-zero to that convention.
 *   - Calls on one context are serialised on that context's HIP stream; the host-pointer
 *     entry points are synchronous on return.  Distinct contexts may be used from distinct
 *     threads/processes (one per GPU).
 *   - There is NO CPU fallback: every compute entry point fails with TSD_E_HIP when no gfx950
 *     device is usable.
 */
#ifndef TSD_H
#define TSD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSD_VERSION 100 /* 0.1.0 */

typedef enum tsd_status {
  TSD_OK = 0,
  TSD_E_ARG = -1,   /* null pointer / bad enum / bad handle */
  TSD_E_SHAPE = -2, /* shape the path does not support (reference: "Returning null matrix") */
  TSD_E_ALLOC = -3,
  TSD_E_HIP = -4, /* HIP runtime error or no usable GPU */
  TSD_E_RCCL = -5,
  TSD_E_STATE = -6, /* call order violated (e.g. forward before weights set) */
  TSD_E_NONFINITE = -7 /* inf / NaN reached a tensor that leaves the device (fp16 activation overflow or non-finite input):
                        * reported by the next synchronisation point, see tsd_debug_nonfinite_count */
} tsd_status;

typedef struct tsd_ctx tsd_ctx;         /* one per GPU: device, stream, workspace arena */
typedef struct tsd_model tsd_model;     /* device-resident packed weights of one model */
typedef struct tsd_session tsd_session; /* device-resident denoise loop state (latents, context, schedule) */

/* ---- introspection ------------------------------------------------------------------- */
int tsd_version(void);
const char* tsd_last_error(void);
int tsd_device_count(void); /* number of visible HIP devices (0 if none / no driver) */

/* ---- context ------------------------------------------------------------------------- */
int tsd_ctx_create(int device, tsd_ctx** out);
int tsd_ctx_destroy(tsd_ctx* ctx);
int tsd_ctx_synchronize(tsd_ctx* ctx);
/* hipEvent timing on the context's own stream (bench.py: torch.cuda.Event would not see it). */
int tsd_ctx_timer_start(tsd_ctx* ctx);
int tsd_ctx_timer_stop(tsd_ctx* ctx, float* elapsed_ms);
/* Per-kernel-class timing with hipEvent pairs around every launch on the context stream (profiling pass
 * only - adds one event pair per launch).  Classes: 0 dense GEMM, 1 conv3x3 implicit GEMM, 2 flash attention,
 * 3 GroupNorm, 4 LayerNorm, 5 tiny-M linear, 6 elementwise/layout, 7 row softmax.  nclass <= 8. */
int tsd_ctx_profile_begin(tsd_ctx* ctx);
int tsd_ctx_profile_end(tsd_ctx* ctx, float* ms_per_class, int* launches_per_class, int nclass);
/* Per-launch records of the current profiling pass (call BEFORE profile_end, which resets):
 * rec[i] = {class, M, N, K, batch} (GEMM-class launches; Sq,Sk,d,B*H for attention), ms[i]; returns the count. */
int tsd_ctx_profile_records(tsd_ctx* ctx, int* rec, float* ms, int cap);

/* ---- op level: one export per reference op struct (host fp32 in/out) ------------------ */

/* `Conv2D.forward` helpers/utils.mojo:1738-1811.  x (C,H,W) [only the first `I` channels are
 * read, :1771], w (O,I,k,k) OIHW (:1718), bias (O) or NULL, symmetric zero padding
 * (pad_h,pad_w) (:1744-1747), stride (:1750-1758).  y (O,Ho,Wo), Ho=(H+2*pad_h-k)/stride_h+1.
 * k in {1,3}. */
int tsd_conv2d_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, const float* w, const float* bias, int I,
                   int O, int k, int pad_h, int pad_w, int stride_h, int stride_w, float* y);

/* `Matrix.pad` helpers/utils.mojo:1383-1413 (zero pad; the encoder's (0,1),(0,1), vae.mojo:115-116). */
int tsd_pad_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int top, int bottom, int left, int right,
                float* y);

/* `GroupNorm.forward` helpers/utils.mojo:1845-1885: y=(x-mu)/(sigma+eps)*gamma over each of
 * `groups` groups of the first `num_channels` channels; population sigma; eps added to sigma
 * (:1871-1873).  y has `num_channels` channels. */
int tsd_groupnorm_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int groups, int num_channels, float eps,
                      float gamma, float* y);

/* `LayerNorm.forward` helpers/utils.mojo:2052-2061, build semantics (SURVEY.md App.A D8):
 * per-row normalisation of x (M,C) with the GroupNorm formula, eps = 1e-5 in the reference. */
int tsd_layernorm_f32(tsd_ctx* ctx, const float* x, int M, int C, float eps, float* y);

/* EXTENSION (not reference behaviour): the norms real, PyTorch-trained checkpoints assume -
 * y = (x - mu) / sqrt(var + eps) * weight[c] + bias[c] (population variance, eps inside the root, per-channel affine;
 * weight / bias may be NULL = ones / zeros), i.e. torch.nn.GroupNorm / LayerNorm.  The reference's GroupNorm has a
 * scalar gamma, an unused beta and eps added to sigma (helpers/utils.mojo:1833-1834,1871-1873); its LayerNorm has no
 * parameters (:2052-2061).  `silu` != 0 fuses x*sigmoid(x) after the GroupNorm. */
int tsd_groupnorm_affine_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, int groups, float eps,
                             const float* weight, const float* bias, int silu, float* y);
int tsd_layernorm_affine_f32(tsd_ctx* ctx, const float* x, int M, int C, float eps, const float* weight,
                             const float* bias, float* y);

/* `SiLU.forward` helpers/utils.mojo:1892-1902 / `Gelu.forward` :1908-1919 (tanh approximation). */
int tsd_silu_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y);
int tsd_gelu_tanh_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y);

/* `Linear.forward` helpers/utils.mojo:1954-1976: y (M,N) = x (M,K) . w(N,K)^T + bias(N)|NULL. */
int tsd_linear_f32(tsd_ctx* ctx, const float* x, int M, int K, const float* w, const float* bias, int N, float* y);

/* `Matrix.matmul` helpers/utils.mojo:1549-1569: c[b] (M,N) = a[b] (M,K) . bmat[b or 0] (K,N),
 * `b_batch` is 1 (broadcast, :770-777) or `batch`. */
int tsd_matmul_f32(tsd_ctx* ctx, const float* a, const float* bmat, int batch, int b_batch, int M, int K, int N,
                   float* c);

/* `Upsample.forward` helpers/utils.mojo:1989-2010, build semantics (App.A D1): nearest x2. */
int tsd_upsample_nearest2x_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, float* y);

/* `Softmax(dim=2)` as used by attention (helpers/utils.mojo:411-448, helpers/attention.mojo:59),
 * build semantics (App.A D6): softmax over the last axis of x (rows, cols). */
int tsd_softmax_lastdim_f32(tsd_ctx* ctx, const float* x, int64_t rows, int cols, float* y);

/* `Self_Attention.forward` helpers/attention.mojo:26-65.  x (T,D); w_in (3D,D), b_in (3D)|NULL;
 * w_out (D,D), b_out (D)|NULL; heads; causal must be 0 on this path (CLIP only). */
int tsd_self_attention_f32(tsd_ctx* ctx, const float* x, int T, int D, int heads, const float* w_in,
                           const float* b_in, const float* w_out, const float* b_out, int causal, float* y);

/* `Cross_Attention.forward` helpers/attention.mojo:96-118.  x (Tq,D), context (Tk,Dc);
 * wq (D,D), wk/wv (D,Dc), wo (D,D); biases (D)|NULL. */
int tsd_cross_attention_f32(tsd_ctx* ctx, const float* x, int Tq, int D, const float* context, int Tk, int Dc,
                            int heads, const float* wq, const float* bq, const float* wk, const float* bk,
                            const float* wv, const float* bv, const float* wo, const float* bo, float* y);

/* `get_time_embedding` helpers/utils.mojo:353-370, build semantics (App.A D9): out[320] =
 * [cos(t f_i), sin(t f_i)], f_i = 10000^(-i/160).  Runs on the device like the rest of the path. */
int tsd_time_embedding_f32(tsd_ctx* ctx, float t, float* out320);

/* ---- block level (host fp32 in/out; weights passed per call, reference field order) ---- */

/* `Time_Embedding.forward` diffusion.mojo:17-21: (320) -> (1280). */
int tsd_time_embedding_mlp_f32(tsd_ctx* ctx, const float* t320, const float* w1, const float* b1,
                               const float* w2, const float* b2, float* out1280);

/* `Unet_Residual_Block.forward` diffusion.mojo:54-72.  x (Cx>=cin,H,W), time (1280).
 * conv1_w (cout,cin,3,3), lin_w (cout,1280), conv2_w (cout,cout,3,3), skip_w (cout,cin,1,1)
 * [used iff cin != cout; may be NULL otherwise].  y (cout,H,W). */
int tsd_unet_residual_block_f32(tsd_ctx* ctx, const float* x, int Cx, int H, int W, const float* time, int cin,
                                int cout, const float* conv1_w, const float* conv1_b, const float* lin_w,
                                const float* lin_b, const float* conv2_w, const float* conv2_b,
                                const float* skip_w, const float* skip_b, float* y);

/* `Unet_Attention_Block.forward` diffusion.mojo:112-147.  x (C,H,W) with C = n_head*n_embed,
 * context (Tk,Dc).  Weights in struct-field order (diffusion.mojo:87-98); `w` is an array of
 * 16 pointers (nw = 16): conv_in_w, conv_in_b, sa_in_w, sa_out_w, sa_out_b, ca_q_w, ca_k_w, ca_v_w,
 * ca_out_w, ca_out_b, geglu1_w, geglu1_b, geglu2_w, geglu2_b, conv_out_w, conv_out_b. */
int tsd_unet_attention_block_f32(tsd_ctx* ctx, const float* x, int n_head, int n_embed, int H, int W,
                                 const float* context, int Tk, int Dc, const float* const* w, int nw, float* y);

/* VAE `Res_Block.forward` vae.mojo:57-67 (GroupNorm 16 groups, no time input). */
int tsd_vae_res_block_f32(tsd_ctx* ctx, const float* x, int H, int W, int cin, int cout, const float* conv1_w,
                          const float* conv1_b, const float* conv2_w, const float* conv2_b, const float* skip_w,
                          const float* skip_b, float* y);

/* VAE `Attention_Block.forward` vae.mojo:17-27 (GroupNorm 32, one head, biases on). */
int tsd_vae_attention_block_f32(tsd_ctx* ctx, const float* x, int C, int H, int W, const float* w_in,
                                const float* b_in, const float* w_out, const float* b_out, float* y);

/* ---- module level: device-resident weights (the measured path) ------------------------ */

/* TSD_MODEL_DIFFUSION_SD15: the full-size (860 M parameter) UNet of BASELINE.json configs[4] - the 12-encoder /
 * bottleneck / 12-decoder layout the reference's 23-layer graph (diffusion.mojo:177-201) was trimmed from, built from
 * the reference's own blocks (Unet_Residual_Block diffusion.mojo:34-72, Unet_Attention_Block :87-147, Upsample
 * :149-160 followed by a 3x3 conv).  It is not defined by the reference: throughput stress configuration only; every
 * entry point that takes a Diffusion accepts it.
 * TSD_MODEL_DIFFUSION_SD15_TORCH: the same graph with the norm semantics of PyTorch-trained SD-1.x checkpoints (extension,
 * see tsd_groupnorm_affine_f32): every GroupNorm / LayerNorm carries per-channel weight and bias parameters (appended
 * after the kind-5 parameter list as `<block>.layerN.weight` / `.bias`, N = the norm's field position in the reference
 * struct), eps sits inside the root, and the output layer's GroupNorm has 32 groups.
 * TSD_MODEL_CLIP_TORCH: the CLIP text encoder (clip.mojo:74-109) with torch LayerNorms - weight and bias of the two
 * LayerNorms of every layer (`playerN.layer1`, `playerN.layer3`) and of the final one (`layernorm`) appended to the
 * kind-4 parameter list; this is exactly Hugging Face's CLIPTextModel (the ViT-L/14 text tower SD-1.x conditions on).
 * TSD_MODEL_DECODER_TORCH / TSD_MODEL_ENCODER_TORCH: the VAE graphs (vae.mojo:94-112,194-219) with the trained VAE's
 * norms - 32 groups in the residual blocks (the reference declares 16, vae.mojo:42-43), per-channel weight / bias of
 * every GroupNorm appended to the kind-2 / kind-3 list (`lN.group_norm1`, `lN.group_norm2`, `lN.group_norm`, and
 * `lN` for the stand-alone norm), eps inside the root: the layout of diffusers' AutoencoderKL decoder / encoder. */
typedef enum tsd_model_kind { TSD_MODEL_DIFFUSION = 1, TSD_MODEL_DECODER = 2, TSD_MODEL_ENCODER = 3, TSD_MODEL_CLIP = 4, TSD_MODEL_DIFFUSION_SD15 = 5, TSD_MODEL_DIFFUSION_SD15_TORCH = 6, TSD_MODEL_CLIP_TORCH = 7, TSD_MODEL_DECODER_TORCH = 8, TSD_MODEL_ENCODER_TORCH = 9 } tsd_model_kind;

/* Parameter inventory in struct-field DFS order: `Diffusion`
 * diffusion.mojo:299-302, `Decoder` vae.mojo:194-219, `Encoder` vae.mojo:94-112.  No GPU needed. */
int tsd_model_param_count(int kind);
int tsd_model_param_info(int kind, int index, char* name, int name_cap, int64_t shape[4], int* ndim, int* used,
                         float* init_bound);

int tsd_model_create(tsd_ctx* ctx, int kind, tsd_model** out);
int tsd_model_destroy(tsd_model* m);
/* Upload one parameter (float32, reference layout: conv OIHW, linear (out,in), bias (out)). */
int tsd_model_set_param(tsd_model* m, int index, const float* data, int64_t numel);
/* Synthetic init on the device (the reference random-initialises in every __init__,
 * helpers/utils.mojo:1716-1727,1938-1945): U(+-init_bound) from the counter RNG
 * value = f(seed, kind*4096+index, element) - bit-identical to tsd/rng.py. */
int tsd_model_init_random(tsd_model* m, uint64_t seed);
/* The packed fp16/fp32 weight blob (one device allocation) - what multi-GPU broadcasts. */
int tsd_model_packed_blob(tsd_model* m, void** device_ptr, size_t* bytes);
int tsd_model_mark_loaded(tsd_model* m); /* after an external write (RCCL broadcast) into the blob */
/* Build the model's derived device buffers now (K-tile-major weight copies, the fused kernels' weight streams, im2col input
 * weights - rebuilt per rank, never broadcast) and wait for them; otherwise the first forward builds them.  Lets a multi-GPU host
 * time that step next to the broadcast (bench.py `derived_buffers_s`).  TSD_E_STATE if a used parameter was never set. */
int tsd_model_prepare(tsd_model* m);

/* `Diffusion.forward` diffusion.mojo:309-318, batched.  latents [B,4,L,L], context [B,T,768],
 * time_emb [B,320] (= get_time_embedding(t) per sample) -> out [B,4,L,L]. */
int tsd_diffusion_forward(tsd_model* m, const float* latents, const float* context, const float* time_emb, int B,
                          int L, int T, float* out);
/* `Decoder.forward` vae.mojo:221-250, batched.  latents [B,4,L,L] -> images [B,3,8L,8L] (raw decoder
 * output; pipeline.mojo:127's rescale is tsd_rescale_images_f32). */
int tsd_decoder_forward(tsd_model* m, const float* latents, int B, int L, float* images);
/* `Encoder.forward` vae.mojo:131-159 (+ metrics_evals :118-129), batched.  images [B,3,S,S] in
 * [-1,1], noise [B,4,S/8,S/8] -> latents [B,4,S/8,S/8]. */
int tsd_encoder_forward(tsd_model* m, const float* images, const float* noise, int B, int S, float* latents);

/* `CLIP.forward` clip.mojo:90-109 (the step before the hot path), intended semantics
 * (App.A D3/D8/D15/D20): tokens [B][T] int32 ids (T <= 77; rows are zero-padded to 77 like clip.mojo:91-93)
 * -> context [B][77][768] fp32, the `context` input of tsd_diffusion_forward / tsd_session_upload.
 * Token embedding + learned position table, 12 x (LayerNorm, causal 12-head self-attention, +res, LayerNorm,
 * Linear 768->3072, quick-GELU x*sigmoid(1.702x), Linear 3072->768, +res), final LayerNorm. */
int tsd_clip_forward(tsd_model* m, const int32_t* tokens, int B, int T, float* context);

/* ---- prompt tokenizer (host only, no GPU): `Tokenizer` + `bpe_encode`, helpers/utils.mojo:229-327, over the
 * tokenizer_clip.bin wire format of tokenizer_creation.py:43-48 (u32 max_token_length; per token f32 score, u32 length,
 * bytes).  pipeline.mojo:37-51: Tokenizer(49408, buf); ids = bpe_encode(prompt.replace(" ", "</w>"), tokenizer). ---- */
typedef struct tsd_tokenizer tsd_tokenizer;
int tsd_tokenizer_create(const char* path, int vocab_size, tsd_tokenizer** out);
int tsd_tokenizer_create_from_memory(const void* data, size_t bytes, int vocab_size, tsd_tokenizer** out);
int tsd_tokenizer_destroy(tsd_tokenizer* t);
/* `Tokenizer.find` :270-287 (with `wrap` :200-209): id of `token`, -1 when absent. */
int tsd_tokenizer_find(const tsd_tokenizer* t, const char* token);
/* vocabulary entry `id` -> bytes (NUL-terminated, truncated to cap) and score */
int tsd_tokenizer_token(const tsd_tokenizer* t, int id, char* out, int cap, float* score);
/* `bpe_encode` :289-327. ids may be NULL to query the count; *complete = 0 when an unknown character ended it early. */
int tsd_tokenizer_encode(const tsd_tokenizer* t, const char* text, int32_t* ids, int cap, int* n_out, int* complete);
/* pipeline.mojo:127 `rescale((-1,1),(0,255),clamp=True)` (helpers/utils.mojo:577-597). */
int tsd_rescale_images_f32(tsd_ctx* ctx, const float* x, int64_t n, float* y);

/* ---- device-resident denoise loop (pipeline.mojo:57-127 + sampler.mojo:15-124) --------- */

/* B samples, latent side L, T context tokens; cfg != 0 runs the UNet on 2B (cond + uncond,
 * SURVEY.md App.A D10).  `decoder` may be NULL when only latents are wanted. */
int tsd_session_create(tsd_model* diffusion, tsd_model* decoder, int B, int L, int T, int cfg, tsd_session** out);
int tsd_session_destroy(tsd_session* s);
/* `DDPMSampler.__init__` + `set_inference_timesteps` sampler.mojo:15-44 (+ `set_strength` :67-73 via
 * start_step: the first `start_step` timesteps are dropped).  beta 0.00085..0.012 scaled-linear. */
int tsd_session_set_schedule(tsd_session* s, int num_training_steps, int num_inference_steps, int start_step);
int tsd_session_num_steps(tsd_session* s);
int tsd_session_timestep(tsd_session* s, int i); /* i-th timestep of the schedule */
/* Upload state.  latents [B,4,L,L]; context [B,T,768]; uncond_context [B,T,768] or NULL;
 * noise [nsteps,B,4,L,L] ~ N(0,1) (an input: App.A D19) or NULL for a noiseless update. */
int tsd_session_upload(tsd_session* s, const float* latents, const float* context, const float* uncond_context,
                       const float* noise, float cfg_scale);
/* Enqueue step i: time embedding -> Diffusion.forward (x1 or x2 with CFG combine) -> DDPMSampler.step
 * (sampler.mojo:75-109).  Asynchronous on the context stream. */
int tsd_session_step(tsd_session* s, int i);
/* `add_noise` sampler.mojo:111-124 at timestep index i (img2img), noise [B,4,L,L] host. */
int tsd_session_add_noise(tsd_session* s, int i, const float* noise);
int tsd_session_decode(tsd_session* s); /* Decoder.forward on the current latents (async) */
int tsd_session_download_latents(tsd_session* s, float* latents);
int tsd_session_download_images(tsd_session* s, int rescale_0_255, float* images);

/* ---- multi-GPU: RCCL over xGMI (one process per GPU) ----------------------------------- */
/* unique_id: 128 bytes from tsd_dist_unique_id on rank 0, shared out-of-band (e.g. torch.distributed). */
int tsd_dist_unique_id(void* id128);
int tsd_dist_init(tsd_ctx* ctx, int rank, int nranks, const void* id128);
int tsd_dist_broadcast_weights(tsd_model* m, int root); /* ncclBroadcast of the packed blob */
/* ranks of the communicator tsd_dist_init created (ncclCommCount): what RCCL itself holds, not what the caller passed in */
int tsd_dist_comm_count(tsd_ctx* ctx, int* nranks);
int tsd_dist_finalize(tsd_ctx* ctx);

/* ---- debug / tuning -------------------------------------------------------------------- */
/* Every switch below belongs to ONE context: the library keeps no process-global mutable state besides the thread-local error
 * string, so two contexts (one per GPU, each driven by its own host thread) may run different settings side by side.  The
 * TSD_* environment variables named here are read once, by tsd_ctx_create, into that context.  A denoise session sizes its
 * workspace for the settings active at upload(): after a tsd_debug_set_* call that CHANGES a setting of its context, step() /
 * decode() fail with TSD_E_STATE until upload() is called again (setting the value that is already there changes nothing).
 * Every tsd_debug_set_* returns the previous setting, always >= 0; an out-of-range value is ignored; the only negative
 * return is TSD_E_ARG (-1) for a NULL context.
 * Non-finite results are sticky per SESSION: once a session download has returned TSD_E_NONFINITE, every later step(), decode()
 * and download of that session returns it again until upload() replaces the session's state. */
/* split-K hand-offs that timed out or paired blocks on different XCDs since the context was created (must be 0);
 * 1 when workgroups map to XCDs round-robin (the precondition of the L2-local split-K hand-off). */
int tsd_debug_splitk_errors(tsd_ctx* ctx);
int tsd_debug_xcd_round_robin(void);
/* Non-finite values (inf / NaN) written to caller-visible tensors on this context since the last report: the path stores
 * activations as fp16 (|x| <= 65504) where the reference computes in fp32 (helpers/utils.mojo:12-15), so an overflow is possible
 * and must never be silent.  Every kernel that produces a tensor that leaves the device counts what it writes; the synchronous
 * entry points, tsd_ctx_synchronize and the session downloads return TSD_E_NONFINITE when the count is non-zero (and clear it).
 * This call reads the count without failing (reset != 0 clears it); < 0 on error.  Synchronises the context's stream. */
int tsd_debug_nonfinite_count(tsd_ctx* ctx, int reset);
/* A/B switch for the fused attention-block kernels of the 64x64 level (kernels_chain.hip): 0 = op-by-op graph, 1 = fused
 * (default; TSD_CHAIN=0 in the environment has the same effect).  Returns the previous setting. */
int tsd_debug_set_fused_attention(tsd_ctx* ctx, int on);
/* Time one GEMM (conv = 0: M = B*H*W, K = Cin) or conv3x3 problem on synthetic device data with tile
 * configuration `cfg` (< 0: dispatcher's choice); average ms per launch over `iters` launches. */
int tsd_debug_gemm_bench(tsd_ctx* ctx, int conv, int B, int H, int W, int Cin, int N, int stride, int ups, int cfg,
                         int iters, float* ms);

/* Run one problem with tile configuration `cfg` and with `ref_cfg`; max |difference| and max |reference|. */
int tsd_debug_gemm_check(tsd_ctx* ctx, int conv, int B, int H, int W, int Cin, int N, int stride, int ups, int cfg,
                         int ref_cfg, float* max_abs_diff, float* max_abs_ref);
/* Same for the fused attention core: Q,K [B][S][H*d], V^T [B][H*d][Sk]. */
int tsd_debug_attn_bench(tsd_ctx* ctx, int B, int H, int d, int Sq, int Sk, int iters, float* ms);
/* The fused attention core runs an optimistic softmax pass (reference fixed after the first key tile) and repeats a
 * workgroup exactly when one of its rows overflowed fp16: number of workgroups that repeated since the last reset
 * (reset != 0 clears the counter); < 0 on error.  Synchronises the context's stream. */
int tsd_debug_attn_exact_passes(tsd_ctx* ctx, int reset);
/* Kernel of the d = 40 attention core: 0 = chosen by shape (default), 1 = 32 queries per wave, 2 = 64 (4-wave workgroups),
 * 3 = the 8-wave two-group kernel (64 queries per wave, 512 per workgroup).
 * All compute every row with the same instruction sequence: bitwise equal results as long as no workgroup takes the exact
 * repeat (there the repeat and the reference moves are decided per workgroup / per wave, i.e. over different row sets).  The
 * default choice depends on the layer shape only, never on the batch.  Returns the previous mode. */
int tsd_debug_set_attn_qb(tsd_ctx* ctx, int mode);
/* Self-attention (Sq == Sk): the optimistic softmax reference is max(row maximum of key tile 0, row maximum over the query's
 * own 32-key block) + headroom; on = 0 restores the tile-0-only reference (to measure what the second reference saves on
 * peaked score distributions).  Returns the previous setting. */
int tsd_debug_set_attn_diag(tsd_ctx* ctx, int on);
/* Residual blocks whose skip path is a 1x1 convolution at the block's own resolution (diffusion.mojo:70-72, vae.mojo:65-67) run it
 * inside the second 3x3 convolution as extra K (on = 1, default); on = 0 runs it as its own GEMM + residual add (the round-2 path,
 * kept for A/B and for the equivalence test).  Returns the previous setting. */
int tsd_debug_set_res_fuse_skip(tsd_ctx* ctx, int on);
/* Self-attention input projection (helpers/attention.mojo:29-31): q | k (token-major) and V^T (channel-major, what the attention
 * kernel reads) come from ONE GEMM over in_proj's 3C rows whose tiles beyond column 2C store transposed (on = 1, default; needs
 * H*W % 32 == 0); on = 0 runs the q/k GEMM and the swapped-operand V^T GEMM as two launches.  Environment: TSD_QKV_FUSE.  Returns
 * the previous setting. */
int tsd_debug_set_qkv_fuse(tsd_ctx* ctx, int on);
/* What this board sustains on the matrix pipe: a register-resident dense fp16 MFMA loop (no LDS, no memory) run for about
 * `ms_target` ms at 4 waves per SIMD; reports the achieved TFLOP/s and the shader clock (GHz) during the run.  The nominal
 * dense peak assumes the boost clock; under matrix-pipe load the board's power limit sets the clock. */
int tsd_debug_mfma_sustained(tsd_ctx* ctx, float ms_target, float* tflops, float* clock_ghz);

/* ---- census -------------------------------------------------------------------------- */
/* Algorithmic GFLOP (2*MAC of conv + linear + attention core) of one forward per sample
 * (SURVEY.md Appendix B): kind, latent side L, context tokens T. */
double tsd_flop_count(int kind, int L, int T);

#ifdef __cplusplus
}
#endif
#endif /* TSD_H */
