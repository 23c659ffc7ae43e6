"""Module-level oracle: UNet / Diffusion / VAE graphs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Single-sample functions in the reference's CHW
layout; `P` is a {name: array} dict keyed as in oracle/spec.py, `prefix` selects the sub-module.
"""
import contextlib

import numpy as np

from . import ops
from .ops import DEFAULT
from .spec import UNET_LAYERS, DECODER_LAYERS, ENCODER_LAYERS, FULL_UNET_STEPS


@contextlib.contextmanager
def using_ops(backend):
    """Run the graphs below on another statement of the ops (oracle.cref.backend(): the reference's loop nests in C)."""
    global ops
    saved, ops = ops, backend
    try:
        yield
    finally:
        ops = saved


def _conv(P, name, x, k_pad, stride=(1, 1), pad_hw=None):
    return ops.conv2d(x, P[name + ".kernel"], P[name + ".bias"], padding=k_pad, stride=stride, pad_hw=pad_hw)


def _lin(P, name, x, use_bias=True):
    return ops.linear(x, P[name + ".weight"], P[name + ".bias"] if use_bias else None)


# ---------------------------------------------------------------------------------------------
# diffusion.mojo


def time_embedding_mlp(P, t320, prefix="time_embed"):
    """`Time_Embedding.forward` diffusion.mojo:17-21: Linear(320,1280) -> SiLU -> Linear(1280,1280)."""
    h = _lin(P, prefix + ".layer1", t320.reshape(1, -1))
    h = ops.silu(h)
    return _lin(P, prefix + ".layer2", h)  # (1, 1280)


def _gn(P, name, x, groups, C, eps, tn):
    """GroupNorm of block field `name`: the reference's, or (tn) torch's with the block's weight / bias."""
    if tn:
        return ops.group_norm_torch(x[:C], groups, eps, P[name + ".weight"], P[name + ".bias"])
    return ops.group_norm(x, groups, C, eps=eps)


def _ln(P, name, x, sem, tn):
    if tn:
        return ops.layer_norm_torch(x, 1e-5, P[name + ".weight"], P[name + ".bias"])
    return ops.layer_norm(x, sem=sem)


def unet_residual_block(P, prefix, x, time, cin, cout, tn=False):
    """`Unet_Residual_Block.forward` diffusion.mojo:54-72.

    GN(32,cin) -> SiLU -> Conv3x3 ; + Linear(1280,cout)(SiLU(time)) broadcast over HxW (:61-65) ;
    GN(32,cout) -> SiLU -> Conv3x3 ; + x (or + Conv1x1(x) when cin != cout, :70-72).
    Consumes only the first `cin` channels of x (App.A D11); `time` is not mutated (App.A D16).
    """
    x = x[:cin]
    residue = x
    h = _gn(P, prefix + ".layer1", x, 32, cin, 1e-5, tn)
    h = ops.silu(h)
    h = _conv(P, prefix + ".layer2", h, (1, 1))
    t = _lin(P, prefix + ".layer3", ops.silu(time))          # (1, cout)
    h = h + t.reshape(cout, 1, 1)
    h = _gn(P, prefix + ".layer4", h, 32, cout, 1e-5, tn)
    h = ops.silu(h)
    h = _conv(P, prefix + ".layer5", h, (1, 1))
    if cin != cout:
        return h + _conv(P, prefix + ".layer6", residue, (0, 0))
    return h + residue


def unet_attention_block(P, prefix, x, context, n_head, n_embed, sem=DEFAULT, tn=False):
    """`Unet_Attention_Block.forward` diffusion.mojo:112-147 (shape walk-through: SURVEY.md App.A.3).

    GN(32, eps 1e-6) -> Conv1x1 -> tokens (HW, C) -> [LN -> self-attn -> +res] ->
    [LN -> cross-attn(context) -> +res] -> [LN -> Linear(C,8C) -> a*gelu(gate) -> Linear(4C,C) -> +res]
    -> (C,H,W) -> Conv1x1 + long residual.
    """
    C = n_head * n_embed
    _, H, W = x.shape
    residue_long = x
    h = _gn(P, prefix + ".layer1", x, 32, C, 1e-6, tn)           # :89,:116
    h = _conv(P, prefix + ".layer2", h, (0, 0))                  # :117
    tok = ops.chw_to_tokens(h)                                   # :118-120
    res = tok
    h = _ln(P, prefix + ".layer3", tok, sem, tn)                 # :122
    h = ops.self_attention(h, n_head, P[prefix + ".layer4.in_proj.weight"], None,
                           P[prefix + ".layer4.out_proj.weight"], P[prefix + ".layer4.out_proj.bias"],
                           sem=sem)                              # :124 (in_bias=False :92)
    tok = h + res                                                # :126
    res = tok
    h = _ln(P, prefix + ".layer5", tok, sem, tn)                 # :129
    h = ops.cross_attention(h, context, n_head,
                            P[prefix + ".layer6.q_proj.weight"], None,
                            P[prefix + ".layer6.k_proj.weight"], None,
                            P[prefix + ".layer6.v_proj.weight"], None,
                            P[prefix + ".layer6.out_proj.weight"], P[prefix + ".layer6.out_proj.bias"],
                            sem=sem)                             # :132 (in_bias=False :94)
    tok = h + res                                                # :133
    res = tok
    h = _ln(P, prefix + ".layer7", tok, sem, tn)                 # :136
    h = _lin(P, prefix + ".layer8", h)                           # :138
    a, gate = np.split(h, 2, axis=-1)                            # chunk(2,2) :138-140
    h = a * (ops.gelu_erf(gate) if tn else ops.gelu_tanh(gate))  # :141 (tn: torch's exact GELU)
    h = _lin(P, prefix + ".layer9", h)                           # :142
    tok = h + res                                                # :143
    h = ops.tokens_to_chw(tok, H, W)                             # :144-145
    return _conv(P, prefix + ".layer10", h, (0, 0)) + residue_long   # :146


def unet(P, x, context, time, sem=DEFAULT, prefix="unet", trace=None):
    """`UNet.forward` diffusion.mojo:228-273: 23 layers, 6 skips, 6 concat(dim 0), 2 upsamples."""
    outs = {}

    def run(i, h):
        kind, a = UNET_LAYERS[i - 1]
        name = f"{prefix}.layer{i}"
        if kind == "conv":
            h = _conv(P, name, h, (1, 1), stride=(a[3], a[3]))
        elif kind == "res":
            h = unet_residual_block(P, name, h, time, *a)
        elif kind == "attn":
            h = unet_attention_block(P, name, h, context, *a, sem=sem)
        elif kind == "up":
            h = ops.upsample_nearest2x(h)
        if trace is not None:
            trace[name] = h
        return h

    h = run(1, x); skip1 = h
    h = run(2, h); h = run(3, h); skip2 = h
    h = run(4, h); skip3 = h
    h = run(5, h); h = run(6, h); skip4 = h
    h = run(7, h); skip5 = h
    h = run(8, h); h = run(9, h); skip6 = h
    h = ops.concat_channels(h, skip6); h = run(10, h); h = run(11, h)      # :253-255
    h = ops.concat_channels(h, skip5); h = run(12, h); h = run(13, h)      # :256-258
    h = run(14, h)
    h = ops.concat_channels(h, skip4); h = run(15, h); h = run(16, h)      # :260-262 (skip4 dead, D11)
    h = ops.concat_channels(h, skip3); h = run(17, h); h = run(18, h)      # :263-265
    h = run(19, h)
    h = ops.concat_channels(h, skip2); h = run(20, h); h = run(21, h)      # :267-269 (skip2 dead, D11)
    h = ops.concat_channels(h, skip1); h = run(22, h); h = run(23, h)      # :270-272
    return h


def unet_full(P, x, context, time, sem=DEFAULT, prefix="unet", trace=None, tn=False):
    """Full-size UNet (spec.FULL_UNET_STEPS; not defined by the reference): every encoder
    output is kept as a skip, every decoder residual block reads concat(x, skip.pop()) on the channel axis."""
    skips = []
    h = x
    for i, (kind, a, flags) in enumerate(FULL_UNET_STEPS, start=1):
        name = f"{prefix}.layer{i}"
        if flags == "pop":
            h = ops.concat_channels(h, skips.pop())
        if kind == "conv":
            h = _conv(P, name, h, (1, 1), stride=(a[3], a[3]))
        elif kind == "upconv":
            h = _conv(P, name, ops.upsample_nearest2x(h), (1, 1))
        elif kind == "res":
            h = unet_residual_block(P, name, h, time, *a, tn=tn)
        elif kind == "attn":
            h = unet_attention_block(P, name, h, context, *a, sem=sem, tn=tn)
        if trace is not None:
            trace[name] = h
        if flags == "push":
            skips.append(h)
    assert not skips
    return h


def diffusion_sd15(P, x, context, t320, sem=DEFAULT, trace=None, tn=False):
    """`Diffusion.forward` (diffusion.mojo:309-318) around the full-size UNet.  tn: PyTorch norm semantics (extension):
    per-channel affine norms, eps inside the root, 32 groups in the output layer."""
    time = time_embedding_mlp(P, t320)
    h = unet_full(P, x, context, time, sem=sem, trace=trace, tn=tn)
    if tn:
        h = ops.silu(ops.group_norm_torch(h, 32, 1e-5, P["final.layer1.weight"], P["final.layer1.bias"]))
        return _conv(P, "final.layer2", h, (1, 1))
    return unet_output_layer(P, h)


def unet_output_layer(P, x, prefix="final"):
    """`UNet_Output_Layer.forward` diffusion.mojo:287-291: GroupNorm(320 groups) -> SiLU -> Conv3x3(320,4)."""
    h = ops.group_norm(x, 320, 320)
    h = ops.silu(h)
    return _conv(P, prefix + ".layer2", h, (1, 1))


def diffusion(P, x, context, t320, sem=DEFAULT, trace=None):
    """`Diffusion.forward` diffusion.mojo:309-318.  x (4,L,L), context (77,768), t320 (320,) -> (4,L,L)."""
    time = time_embedding_mlp(P, t320)
    h = unet(P, x, context, time, sem=sem, trace=trace)
    return unet_output_layer(P, h)


# ---------------------------------------------------------------------------------------------
# vae.mojo


def vae_attention_block(P, prefix, x, sem=DEFAULT, tn=False):
    """`Attention_Block.forward` vae.mojo:17-27: GN32 -> tokens -> Self_Attention(1 head, biases on) -> + x."""
    C, H, W = x.shape
    h = _gn(P, prefix + ".group_norm", x, 32, C, 1e-6 if tn else 1e-5, tn)  # trained VAE (diffusers AutoencoderKL): eps 1e-6
    tok = ops.chw_to_tokens(h)
    tok = ops.self_attention(tok, 1, P[prefix + ".attention.in_proj.weight"], P[prefix + ".attention.in_proj.bias"],
                             P[prefix + ".attention.out_proj.weight"], P[prefix + ".attention.out_proj.bias"], sem=sem)
    return ops.tokens_to_chw(tok, H, W) + x


def vae_res_block(P, prefix, x, cin, cout, tn=False):
    """`Res_Block.forward` vae.mojo:57-67: GN16 -> SiLU -> Conv3x3 -> GN16 -> SiLU -> Conv3x3 ; + x or + Conv1x1(x)."""
    g = 32 if tn else 16  # the trained VAE has 32 groups (extension); the reference declares 16 (vae.mojo:42-43)
    h = _gn(P, prefix + ".group_norm1", x, g, cin, 1e-6 if tn else 1e-5, tn)
    h = ops.silu(h)
    h = _conv(P, prefix + ".conv1", h, (1, 1))
    h = _gn(P, prefix + ".group_norm2", h, g, cout, 1e-6 if tn else 1e-5, tn)
    h = ops.silu(h)
    h = _conv(P, prefix + ".conv2", h, (1, 1))
    res = x if cin == cout else _conv(P, prefix + ".res_conv_layer", x, (0, 0))
    return h + res


def _vae_run(P, layers, x, sem, trace=None, tn=False):
    h = x
    for i, (kind, a) in enumerate(layers, start=1):
        name = f"l{i}"
        if kind == "conv":
            p = (a[2] // 2, a[2] // 2)
            h = _conv(P, name, h, p)
        elif kind == "conv_s2":      # vae.mojo:138-139: pad (0,1),(0,1) then stride-2 conv without padding
            h = _conv(P, name, h, (0, 0), stride=(2, 2), pad_hw=((0, 1), (0, 1)))
        elif kind == "res":
            h = vae_res_block(P, name, h, *a, tn=tn)
        elif kind == "attn":
            h = vae_attention_block(P, name, h, sem, tn=tn)
        elif kind == "up":
            h = ops.upsample_nearest2x(h)
        elif kind == "gn":
            h = _gn(P, name, h, a[0], a[1], 1e-6 if tn else 1e-5, tn)
        elif kind == "silu":
            h = ops.silu(h)
        if trace is not None:
            trace[name] = h
    return h


def decoder(P, x, sem=DEFAULT, trace=None, tn=False):
    """`Decoder.forward` vae.mojo:221-250: x/0.18215 then 26 layers; (4,L,L) -> (3,8L,8L).  Pure (App.A D14)."""
    return _vae_run(P, DECODER_LAYERS, x / x.dtype.type(0.18215), sem, trace, tn=tn)


def encoder(P, x, noise, sem=DEFAULT, trace=None, tn=False):
    """`Encoder.forward` vae.mojo:131-159 + `metrics_evals` :118-129.

    (3,S,S) -> (8,S/8,S/8) -> mean, logvar = chunk(0,2); logvar clamp(-30,20);
    out = (mean + noise*exp(0.5*logvar)) * 0.18215."""
    h = _vae_run(P, ENCODER_LAYERS, x, sem, trace, tn=tn)
    mean, logvar = np.split(h, 2, axis=0)
    logvar = np.clip(logvar, -30.0, 20.0)
    std = np.sqrt(np.exp(logvar))
    return (mean + noise * std) * x.dtype.type(0.18215)


# ---------------------------------------------------------------------------------------------
# clip.mojo (the step before the hot path; intended semantics only, App.A D3/D8/D15/D20)

def clip_layer(P, prefix, x, n_head=12, sem=DEFAULT, tn=False):
    """`ClipPlayer.forward` clip.mojo:36-53: LN -> causal self-attention -> +res -> LN -> Linear -> quick-GELU -> Linear -> +res."""
    res = x
    h = _ln(P, prefix + ".layer1", x, sem, tn)
    h = ops.self_attention(h, n_head, P[prefix + ".layer2.in_proj.weight"], P[prefix + ".layer2.in_proj.bias"],
                           P[prefix + ".layer2.out_proj.weight"], P[prefix + ".layer2.out_proj.bias"], causal=True, sem=sem)
    x = h + res
    res = x
    h = _ln(P, prefix + ".layer3", x, sem, tn)
    h = _lin(P, prefix + ".layer4", h)
    h = ops.quick_gelu(h)
    h = _lin(P, prefix + ".layer5", h)
    return h + res


def clip(P, tokens, sem=DEFAULT, n_token=77, tn=False):
    """`CLIP.forward` clip.mojo:90-109: token ids (<= 77, zero-padded to 77 like :91-93) -> (77, 768)."""
    t = np.zeros(n_token, dtype=np.int64)
    tokens = np.asarray(tokens, dtype=np.int64).reshape(-1)
    t[: len(tokens)] = tokens
    x = ops.embedding(t, P["embedding.token.weight"]) + P["embedding.position"].reshape(n_token, -1)  # clip.mojo:17-20
    for i in range(1, 13):
        x = clip_layer(P, f"player{i}", x, sem=sem, tn=tn)
    return _ln(P, "layernorm", x, sem, tn)
