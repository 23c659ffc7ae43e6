"""Pins of the CPU oracle (runs without a GPU).

The reference ships no tests or golden vectors, so the oracle is pinned here
against torch.nn.functional for the ops whose formula equals PyTorch's, and against hand-derived
values for the formulas that deliberately differ (GroupNorm sigma+eps, no-affine LayerNorm,
tanh-GELU, DDPM schedule constants)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models, ops, sampler, spec
from util import randn

T = torch.tensor


@pytest.mark.parametrize("stride,pad", [((1, 1), (1, 1)), ((2, 2), (1, 1)), ((2, 2), (0, 0)), ((1, 1), (0, 0))])
def test_conv2d_matches_torch(stride, pad):
    x, w, b = randn(1, 6, 9, 11), randn(2, 5, 6, 3, 3), randn(3, 5)
    y = ops.conv2d(x, w, b, padding=pad, stride=stride)
    yt = F.conv2d(T(x)[None], T(w), T(b), stride=stride, padding=pad)[0].numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-5, atol=1e-5)


def test_conv2d_1x1_and_first_channels_only():
    x, w = randn(4, 10, 5, 5), randn(5, 3, 6, 1, 1)
    y = ops.conv2d(x, w, None)  # reads only the first 6 channels (helpers/utils.mojo:1771)
    yt = F.conv2d(T(x[:6])[None], T(w))[0].numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-5, atol=1e-5)


def test_conv2d_encoder_asymmetric_pad():
    x, w, b = randn(6, 4, 8, 8), randn(7, 4, 4, 3, 3), randn(8, 4)
    y = ops.conv2d(x, w, b, stride=(2, 2), pad_hw=((0, 1), (0, 1)))  # vae.mojo:115-116,138-139
    yt = F.conv2d(F.pad(T(x)[None], (0, 1, 0, 1)), T(w), T(b), stride=2)[0].numpy()
    assert y.shape == (4, 4, 4)
    np.testing.assert_allclose(y, yt, rtol=1e-5, atol=1e-5)


def test_conv2d_chunked_equals_unchunked():
    x, w = randn(9, 8, 12, 12), randn(10, 4, 8, 3, 3)
    a = ops.conv2d(x, w, None, padding=(1, 1))
    b = ops.conv2d(x, w, None, padding=(1, 1), max_cols_bytes=4096)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)


def test_elementwise_match_torch():
    x = randn(11, 4, 33)
    np.testing.assert_allclose(ops.silu(x), F.silu(T(x)).numpy(), rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ops.gelu_tanh(x), F.gelu(T(x), approximate="tanh").numpy(), rtol=1e-5, atol=1e-6)
    x3 = randn(12, 4, 3, 5)
    np.testing.assert_allclose(ops.upsample_nearest2x(x3),
                               F.interpolate(T(x3)[None], scale_factor=2, mode="nearest")[0].numpy())
    np.testing.assert_allclose(ops.pad(x3, (0, 1), (2, 0)), F.pad(T(x3), (2, 0, 0, 1)).numpy())


def test_linear_and_attention_match_torch():
    t, wi, wo, bo = randn(12, 10, 16), randn(13, 48, 16), randn(14, 16, 16), randn(15, 16)
    y = ops.self_attention(t, 4, wi, None, wo, bo)
    qkv = T(t) @ T(wi).T
    q, k, v = qkv.chunk(3, -1)
    sh = lambda a: a.view(10, 4, 4).transpose(0, 1)[None]  # noqa: E731
    o = F.scaled_dot_product_attention(sh(q), sh(k), sh(v))[0].transpose(0, 1).reshape(10, 16)
    np.testing.assert_allclose(y, (o @ T(wo).T + T(bo)).numpy(), rtol=1e-4, atol=1e-5)
    ctx, wk, wv = randn(16, 7, 12), randn(17, 16, 12), randn(18, 16, 12)
    wq = randn(19, 16, 16)
    y = ops.cross_attention(t, ctx, 2, wq, None, wk, None, wv, None, wo, bo)
    q, k, v = T(t) @ T(wq).T, T(ctx) @ T(wk).T, T(ctx) @ T(wv).T
    shq = lambda a, n: a.view(n, 2, 8).transpose(0, 1)[None]  # noqa: E731
    o = F.scaled_dot_product_attention(shq(q, 10), shq(k, 7), shq(v, 7))[0].transpose(0, 1).reshape(10, 16)
    np.testing.assert_allclose(y, (o @ T(wo).T + T(bo)).numpy(), rtol=1e-4, atol=1e-5)


def test_groupnorm_hand_derived():
    # one group of 2 channels x 1 x 2: values 1,2,3,6 -> mu=3, population sigma=sqrt(3.5); eps ADDED to sigma
    x = np.array([[[1.0, 2.0]], [[3.0, 6.0]]], dtype=np.float32)
    y = ops.group_norm(x, 1, 2, eps=0.5)
    exp = (x - 3.0) / (np.sqrt(3.5) + 0.5)
    np.testing.assert_allclose(y, exp, rtol=1e-6)
    # differs from torch's sqrt(var+eps) when eps is large, equals it (up to eps placement) when tiny
    yt = F.group_norm(T(x)[None], 1, eps=0.0)[0].numpy()
    np.testing.assert_allclose(ops.group_norm(x, 1, 2, eps=0.0), yt, rtol=1e-5, atol=1e-6)
    # only the first num_channels channels are normalised/returned (helpers/utils.mojo:1847,1857-1859)
    x3 = randn(20, 6, 3, 3)
    np.testing.assert_allclose(ops.group_norm(x3, 2, 4), ops.group_norm(x3[:4], 2, 4))


def test_groupnorm_vs_torch_small_eps():
    x = randn(21, 32, 5, 7)
    y = ops.group_norm(x, 8, 32, eps=1e-5)
    yt = F.group_norm(T(x)[None], 8, eps=0.0)[0].numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-3, atol=1e-4)


def test_torch_style_norm_extension_matches_torch():
    """The f-4 extension (per-channel affine, eps inside the root) IS torch's GroupNorm / LayerNorm."""
    x = randn(61, 32, 5, 7) * 2.0 + 0.5
    w, b = 1.0 + 0.3 * randn(62, 32), 0.2 * randn(63, 32)
    for eps in (1e-5, 1e-2):
        yt = F.group_norm(T(x)[None], 8, T(w), T(b), eps=eps)[0].numpy()
        np.testing.assert_allclose(ops.group_norm_torch(x, 8, eps, w, b), yt, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ops.group_norm_torch(x, 8, 1e-5), F.group_norm(T(x)[None], 8, eps=1e-5)[0].numpy(),
                               rtol=1e-5, atol=1e-5)
    t = randn(64, 5, 16) * 3.0
    lw, lb = 1.0 + 0.3 * randn(65, 16), 0.2 * randn(66, 16)
    np.testing.assert_allclose(ops.layer_norm_torch(t, 1e-5, lw, lb), F.layer_norm(T(t), (16,), T(lw), T(lb), eps=1e-5).numpy(),
                               rtol=1e-5, atol=1e-5)
    g = randn(67, 4, 50) * 3.0
    np.testing.assert_allclose(ops.gelu_erf(g), F.gelu(T(g)).numpy(), rtol=1e-5, atol=1e-6)
    assert np.abs(ops.gelu_erf(g) - ops.gelu_tanh(g)).max() > 1e-4  # the reference's tanh form is a different function
    # and it differs from the reference's formula exactly where the two put eps
    assert not np.allclose(ops.group_norm_torch(x, 8, 1e-2), ops.group_norm(x, 8, 32, eps=1e-2), atol=1e-4)


def test_layernorm_per_token_and_literal_global():
    x = randn(22, 5, 16)
    y = ops.layer_norm(x)
    yt = F.layer_norm(T(x), (16,), eps=0.0).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-3, atol=1e-4)
    g = ops.layer_norm(x, sem=ops.Semantics(literal_layernorm_global=True))
    assert abs(g.mean()) < 1e-5 and not np.allclose(g, y)


def test_literal_quirk_flags_documented():
    """Deterministic literal behaviours of SURVEY.md Appendix A (documentation only, never graded on GPU)."""
    e = ops.time_embedding(500.0, sem=ops.Semantics(literal_time_freqs=True))
    np.testing.assert_allclose(e[:160], 1.0, atol=1e-6)   # D9: freqs underflow -> cos(0)=1
    np.testing.assert_allclose(e[160:], 0.0, atol=1e-6)
    s = randn(23, 2, 3, 4)
    lit = ops.softmax_lastdim(s, ops.Semantics(literal_softmax_axis=True))
    np.testing.assert_allclose(lit.sum(axis=-2), 1.0, rtol=1e-5)  # D6: normalised over queries
    np.testing.assert_allclose(ops.softmax_lastdim(s).sum(axis=-1), 1.0, rtol=1e-5)
    x = randn(24, 6, 8)
    a = ops._split_heads(x, 2, ops.Semantics(literal_head_split=True))  # D5 flat reinterpretation
    assert a.shape == (2, 6, 4) and not np.allclose(a, ops._split_heads(x, 2, ops.DEFAULT))
    np.testing.assert_allclose(ops._split_heads(x, 1, ops.Semantics(literal_head_split=True)), ops._split_heads(x, 1, ops.DEFAULT))


def test_time_embedding_formula():
    e = ops.time_embedding(10.0)
    f = 10000.0 ** (-np.arange(160) / 160.0)
    np.testing.assert_allclose(e[:160], np.cos(10.0 * f), atol=2e-6)
    np.testing.assert_allclose(e[160:], np.sin(10.0 * f), atol=2e-6)


def test_ddpm_schedule_constants():
    s = sampler.DDPMSampler(1000)
    assert abs(s.alphas_cumprod[0] - 0.99915) < 1e-6
    assert abs(s.alphas_cumprod[999] - 0.004660) < 1e-6
    s.set_inference_timesteps(50)
    assert list(s.timesteps[:3]) == [980, 960, 940] and s.timesteps[-1] == 0 and len(s.timesteps) == 50
    s.set_strength(0.6)                                        # 30 steps remain (BASELINE config 4)
    assert len(s.timesteps) == 30 and s.start_step == 20
    s10 = sampler.DDPMSampler(1000)
    s10.set_inference_timesteps(10)
    assert list(s10.timesteps) == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]  # BASELINE config 1


def test_ddpm_step_matches_closed_form():
    s = sampler.DDPMSampler(1000)
    s.set_inference_timesteps(50)
    x, eps, z = randn(25, 4, 8, 8), randn(26, 4, 8, 8), randn(27, 4, 8, 8)
    t = 500
    a_t, a_p = float(s.alphas_cumprod[t]), float(s.alphas_cumprod[t - 20])
    x0 = (x - np.sqrt(1 - a_t) * eps) / np.sqrt(a_t)
    cur_a = a_t / a_p
    mean = np.sqrt(a_p) * (1 - cur_a) / (1 - a_t) * x0 + np.sqrt(cur_a) * (1 - a_p) / (1 - a_t) * x
    var = (1 - a_p) / (1 - a_t) * (1 - cur_a)
    np.testing.assert_allclose(s.step(t, x, eps, z), mean + np.sqrt(var) * z, rtol=1e-4, atol=1e-5)
    # last step: no noise, alpha_prev = 1 -> x_prev = x0
    a0 = float(s.alphas_cumprod[0])
    np.testing.assert_allclose(s.step(0, x, eps, z), (x - np.sqrt(1 - a0) * eps) / np.sqrt(a0), rtol=1e-4, atol=1e-4)


def test_parameter_census_matches_survey():
    n = lambda ps: sum(p.numel for p in ps if p.used)  # noqa: E731
    assert abs(n(spec.diffusion_params()) / 1e6 - 299.74) < 0.01
    assert abs(n(spec.decoder_params()) / 1e6 - 49.47) < 0.01
    assert abs(n(spec.encoder_params()) / 1e6 - 34.15) < 0.01


def test_dead_concat_equals_literal_truncation():
    """App.A D11: a residual block declared with fewer in_channels than the concat ignores the tail."""
    P = {}
    for p in [q for q in spec.diffusion_params() if q.name.startswith("unet.layer20.")]:
        P[p.name] = randn(hash(p.name) % 1000 + 100, *p.shape) * 0.05
    x, skip, time = randn(28, 640, 4, 4), randn(29, 320, 4, 4), randn(30, 1, 1280)
    a = models.unet_residual_block(P, "unet.layer20", np.concatenate([x, skip]), time, 640, 320)
    b = models.unet_residual_block(P, "unet.layer20", x, time, 640, 320)
    np.testing.assert_array_equal(a, b)


def test_cfg_combine_and_rescale():
    c, u = randn(31, 4, 4, 4), randn(32, 4, 4, 4)
    np.testing.assert_allclose(sampler.cfg_combine(c, u, 7.5), (c - u) * 7.5 + u, rtol=1e-6)
    np.testing.assert_allclose(ops.rescale_to_u8_range(np.array([-2.0, -1.0, 0.0, 1.0, 3.0], dtype=np.float32)),
                               [0.0, 0.0, 127.5, 255.0, 255.0])


def test_clip_pieces_match_torch():
    """CLIP: quick-GELU, the causal mask, the embedding gather and one ClipPlayer layer
    against torch restatements of the intended semantics (App.A D3 / D7 / D15)."""
    x = randn(40, 5, 33)
    np.testing.assert_allclose(ops.quick_gelu(x), (T(x) * torch.sigmoid(1.702 * T(x))).numpy(), rtol=1e-5, atol=1e-6)
    table = randn(41, 11, 8)
    np.testing.assert_array_equal(ops.embedding([3, 0, 10, 3], table), table[[3, 0, 10, 3]])
    # causal self-attention: F.scaled_dot_product_attention(is_causal=True)
    t, wi, bi, wo, bo = randn(42, 9, 24), randn(43, 72, 24) * 0.2, randn(44, 72) * 0.1, randn(45, 24, 24) * 0.2, randn(46, 24)
    y = ops.self_attention(t, 3, wi, bi, wo, bo, causal=True)
    qkv = T(t) @ T(wi).T + T(bi)
    q, k, v = qkv.chunk(3, -1)
    sh = lambda a: a.view(9, 3, 8).transpose(0, 1)[None]  # noqa: E731
    o = F.scaled_dot_product_attention(sh(q), sh(k), sh(v), is_causal=True)[0].transpose(0, 1).reshape(9, 24)
    np.testing.assert_allclose(y, (o @ T(wo).T + T(bo)).numpy(), rtol=1e-4, atol=1e-5)
    # first token attends only to itself: its output is out_proj(v_0)
    np.testing.assert_allclose(y[0], (v[0] @ T(wo).T + T(bo)).numpy(), rtol=1e-4, atol=1e-5)


def test_clip_parameter_census_and_padding():
    n = sum(p.numel for p in spec.clip_params())
    assert n == 49408 * 768 + 77 * 768 + 12 * (3 * 768 * 768 + 3 * 768 + 768 * 768 + 768 + 2 * 4 * 768 * 768 + 4 * 768 + 768)
    P = {p.name: (randn(hash(p.name) % 997 + 200, *p.shape) * 0.05).astype(np.float32) for p in spec.clip_params()
         if not p.name.startswith("embedding.token")}
    P["embedding.token.weight"] = randn(47, 64, 768)[np.arange(49408) % 64]  # small table tiled to the vocabulary size
    a = models.clip(P, [5, 9, 2])
    b = models.clip(P, [5, 9, 2] + [0] * 74)  # prompts are zero-padded to 77 ids (clip.mojo:91-93)
    assert a.shape == (77, 768)
    np.testing.assert_array_equal(a, b)
    # causal: the first three output rows do not depend on later tokens
    c = models.clip(P, [5, 9, 2, 40, 41])
    np.testing.assert_allclose(a[:3], c[:3], rtol=1e-5, atol=1e-6)
    # per-token LayerNorm with no affine at the end: zero mean over channels
    np.testing.assert_allclose(a.mean(axis=-1), 0.0, atol=1e-4)


def test_clip_graph_matches_transformers_cliptextmodel():
    """Independent pin of the CLIP text encoder (f-3): the oracle's intended-semantics graph (clip.mojo:36-109) against
    Hugging Face's CLIPTextModel - the architecture clip.mojo was written from - on the same random weights: token +
    position embedding, 12 pre-LN layers with causal 12-head attention and quick-GELU, final LayerNorm.  The model's
    LayerNorm weights stay (1, 0) (the reference's LayerNorm has no parameters) and activations are O(1), so where the
    two put eps (sigma + eps vs sqrt(var + eps)) moves the result by ~1e-5 only."""
    tr = pytest.importorskip("transformers")
    torch.manual_seed(0)
    cfg = tr.CLIPTextConfig(vocab_size=1000, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                            num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5,
                            eos_token_id=999, bos_token_id=998, pad_token_id=0)
    m = tr.CLIPTextModel(cfg).eval()
    sd = m.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if "layer_norm" in k:
                continue
            if "embedding" in k:
                v.normal_(0, 1.0)
            elif k.endswith("weight"):
                v.normal_(0, 1.0 / np.sqrt(v.shape[1]))
            else:
                v.normal_(0, 0.1)
    pre = "text_model." if any(k.startswith("text_model.") for k in sd) else ""
    g = lambda k: sd[pre + k].numpy()  # noqa: E731
    P = {"embedding.token.weight": g("embeddings.token_embedding.weight"),
         "embedding.position": g("embeddings.position_embedding.weight").reshape(-1)}
    for i in range(12):
        h, n = f"encoder.layers.{i}.", f"player{i + 1}"
        P[n + ".layer2.in_proj.weight"] = np.concatenate([g(h + f"self_attn.{x}_proj.weight") for x in "qkv"])
        P[n + ".layer2.in_proj.bias"] = np.concatenate([g(h + f"self_attn.{x}_proj.bias") for x in "qkv"])
        P[n + ".layer2.out_proj.weight"], P[n + ".layer2.out_proj.bias"] = g(h + "self_attn.out_proj.weight"), g(h + "self_attn.out_proj.bias")
        P[n + ".layer4.weight"], P[n + ".layer4.bias"] = g(h + "mlp.fc1.weight"), g(h + "mlp.fc1.bias")
        P[n + ".layer5.weight"], P[n + ".layer5.bias"] = g(h + "mlp.fc2.weight"), g(h + "mlp.fc2.bias")
    tok = np.random.RandomState(1).randint(1, 998, size=77)
    with torch.no_grad():
        ref = m(input_ids=torch.from_numpy(tok)[None]).last_hidden_state[0].numpy()
    y = models.clip(P, tok)
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 1e-4
