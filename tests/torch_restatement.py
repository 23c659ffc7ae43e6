"""A SECOND, independently written restatement of the reference's module graphs - test infrastructure only.

`Diffusion.forward` (/root/reference/diffusion.mojo:309-318 -> UNet.forward :228-273, Unet_Residual_Block.forward :54-72,
Unet_Attention_Block.forward :112-147, UNet_Output_Layer.forward :287-291, Time_Embedding.forward :17-21) and
`Decoder.forward` (/root/reference/vae.mojo:221-250 -> Res_Block.forward :57-67, Attention_Block.forward :17-27), with the
attention of /root/reference/helpers/attention.mojo:26-65 and :96-118, written directly from those files with
torch.nn.functional in float64.  It deliberately does NOT import or call anything from `oracle/`: the numpy oracle and this
file are two independent readings of the same Mojo source, and tests/test_oracle_second_statement.py requires them to agree.

Where the literal Mojo is ill-defined the same "build implements" semantics as SURVEY.md Appendix A are used (the reference
cannot run, so these are part of the contract, not of either restatement):
  * GroupNorm / LayerNorm: (x - mean) / (sigma + eps) with the POPULATION sigma and eps added to sigma, no affine
    (helpers/utils.mojo:1845-1885, :1372-1380); LayerNorm is per token (App.A D8);
  * a block declared with fewer input channels than the concat provides reads the first `in_channels` (App.A D11);
  * Upsample is nearest-neighbour x2 whatever its argument (App.A D1); heads are split the standard way and the softmax
    runs over the keys (App.A D5, D6); Gelu is the tanh form (helpers/utils.mojo:1914).
Parameters come in as a {name: array} dict in the reference's own layouts (conv OIHW, linear (out, in))."""
import math

import torch
import torch.nn.functional as F

DT = torch.float64


def _t(a):
    return torch.as_tensor(a, dtype=DT)


def _norm_groups(x, groups, channels, eps):
    """x (C, H, W): the first `channels` channels in `groups` groups, (x - mu) / (sigma_population + eps)."""
    x = x[:channels]
    c, h, w = x.shape
    g = x.reshape(groups, -1)
    mu = g.mean(dim=1, keepdim=True)
    sigma = ((g - mu) ** 2).mean(dim=1, keepdim=True).sqrt()
    return ((g - mu) / (sigma + eps)).reshape(c, h, w)


def _norm_tokens(x, eps=1e-5):
    """x (T, C): every token normalised over its C features."""
    mu = x.mean(dim=1, keepdim=True)
    sigma = ((x - mu) ** 2).mean(dim=1, keepdim=True).sqrt()
    return (x - mu) / (sigma + eps)


def _conv(P, name, x, pad, stride=1):
    return F.conv2d(x[None], _t(P[name + ".kernel"]), _t(P[name + ".bias"]), stride=stride, padding=pad)[0]


def _lin(P, name, x, bias=True):
    return F.linear(x, _t(P[name + ".weight"]), _t(P[name + ".bias"]) if bias else None)


def _heads(x, n):  # (T, C) -> (n, T, C/n)
    t, c = x.shape
    return x.reshape(t, n, c // n).permute(1, 0, 2)


def _attend(q, k, v, n):
    """softmax_keys(q k^T / sqrt(d_head)) v per head, heads merged back to (T, C)  (attention.mojo:46-62, :108-115)."""
    qh, kh, vh = _heads(q, n), _heads(k, n), _heads(v, n)
    w = torch.matmul(qh, kh.transpose(1, 2)) / math.sqrt(q.shape[1] // n)
    w = torch.softmax(w, dim=2)
    o = torch.matmul(w, vh)  # (n, T, d)
    return o.permute(1, 0, 2).reshape(q.shape[0], q.shape[1])


def self_attention(P, name, x, n, in_bias, out_bias=True):
    qkv = _lin(P, name + ".in_proj", x, in_bias)
    q, k, v = qkv.chunk(3, dim=1)
    return _lin(P, name + ".out_proj", _attend(q, k, v, n), out_bias)


def cross_attention(P, name, x, ctx, n):
    q = _lin(P, name + ".q_proj", x, False)
    k = _lin(P, name + ".k_proj", ctx, False)
    v = _lin(P, name + ".v_proj", ctx, False)
    return _lin(P, name + ".out_proj", _attend(q, k, v, n), True)


def res_block(P, name, x, time, cin, cout):  # diffusion.mojo:54-72
    x = x[:cin]
    h = F.silu(_norm_groups(x, 32, cin, 1e-5))
    h = _conv(P, name + ".layer2", h, 1)
    t = _lin(P, name + ".layer3", F.silu(time))
    h = h + t.reshape(cout, 1, 1)
    h = F.silu(_norm_groups(h, 32, cout, 1e-5))
    h = _conv(P, name + ".layer5", h, 1)
    return h + (_conv(P, name + ".layer6", x, 0) if cin != cout else x)


def attention_block(P, name, x, ctx, n_head, n_embed):  # diffusion.mojo:112-147
    c = n_head * n_embed
    _, hh, ww = x.shape
    h = _norm_groups(x, 32, c, 1e-6)
    h = _conv(P, name + ".layer2", h, 0)
    tok = h.reshape(c, hh * ww).t()  # (HW, C), row-major over (H, W)
    tok = tok + self_attention(P, name + ".layer4", _norm_tokens(tok), n_head, in_bias=False)
    tok = tok + cross_attention(P, name + ".layer6", _norm_tokens(tok), ctx, n_head)
    a, gate = _lin(P, name + ".layer8", _norm_tokens(tok)).chunk(2, dim=1)
    tok = tok + _lin(P, name + ".layer9", a * F.gelu(gate, approximate="tanh"))
    h = tok.t().reshape(c, hh, ww)
    return _conv(P, name + ".layer10", h, 0) + x


def _up(x):
    return F.interpolate(x[None], scale_factor=2, mode="nearest")[0]


def diffusion(P, latents, context, time_emb):
    """latents (4, L, L), context (T, 768), time_emb (320,) -> eps (4, L, L)."""
    x, ctx, t = _t(latents), _t(context), _t(time_emb).reshape(1, 320)
    time = _lin(P, "time_embed.layer2", F.silu(_lin(P, "time_embed.layer1", t)))[0]  # (1280,)
    R = lambda i, h, cin, cout: res_block(P, f"unet.layer{i}", h, time, cin, cout)
    A = lambda i, h, d: attention_block(P, f"unet.layer{i}", h, ctx, 8, d)
    h = _conv(P, "unet.layer1", x, 1); s1 = h
    h = A(3, R(2, h, 320, 320), 40); s2 = h
    h = _conv(P, "unet.layer4", h, 1, 2); s3 = h
    h = A(6, R(5, h, 320, 640), 80); s4 = h
    h = _conv(P, "unet.layer7", h, 1, 2); s5 = h
    h = A(9, R(8, h, 640, 1280), 160); s6 = h
    h = A(11, R(10, torch.cat([h, s6]), 2560, 1280), 160)
    h = A(13, R(12, torch.cat([h, s5]), 1920, 1280), 160)
    h = _up(h)
    h = A(16, R(15, torch.cat([h, s4]), 1280, 640), 80)   # declared 1280 inputs: the skip part is never read
    h = A(18, R(17, torch.cat([h, s3]), 960, 640), 80)
    h = _up(h)
    h = A(21, R(20, torch.cat([h, s2]), 640, 320), 40)    # declared 640 inputs
    h = A(23, R(22, torch.cat([h, s1]), 640, 320), 40)
    h = F.silu(_norm_groups(h, 320, 320, 1e-5))            # UNet_Output_Layer: GroupNorm(320 groups)
    return _conv(P, "final.layer2", h, 1).numpy()


def vae_res(P, name, x, cin, cout):  # vae.mojo:57-67
    h = _conv(P, name + ".conv1", F.silu(_norm_groups(x, 16, cin, 1e-5)), 1)
    h = _conv(P, name + ".conv2", F.silu(_norm_groups(h, 16, cout, 1e-5)), 1)
    return h + (_conv(P, name + ".res_conv_layer", x, 0) if cin != cout else x)


def vae_attention(P, name, x):  # vae.mojo:17-27
    c, hh, ww = x.shape
    tok = _norm_groups(x, 32, c, 1e-5).reshape(c, hh * ww).t()
    tok = self_attention(P, name + ".attention", tok, 1, in_bias=True)
    return tok.t().reshape(c, hh, ww) + x


def decoder(P, latents):
    """latents (4, L, L) -> image (3, 8L, 8L) before the [0, 255] rescale (vae.mojo:221-250)."""
    h = _t(latents) / 0.18215
    h = _conv(P, "l2", _conv(P, "l1", h, 0), 1)
    h = vae_attention(P, "l4", vae_res(P, "l3", h, 512, 512))
    for i in (5, 6, 7, 8):
        h = vae_res(P, f"l{i}", h, 512, 512)
    h = _conv(P, "l10", _up(h), 1)
    for i in (11, 12, 13):
        h = vae_res(P, f"l{i}", h, 512, 512)
    h = _conv(P, "l15", _up(h), 1)
    h = vae_res(P, "l16", h, 512, 256)
    h = vae_res(P, "l18", vae_res(P, "l17", h, 256, 256), 256, 256)
    h = _conv(P, "l20", _up(h), 1)
    h = vae_res(P, "l21", h, 256, 128)
    h = vae_res(P, "l23", vae_res(P, "l22", h, 128, 128), 128, 128)
    return _conv(P, "l26", F.silu(_norm_groups(h, 32, 128, 1e-5)), 1).numpy()
