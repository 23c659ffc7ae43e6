"""Op-level oracle: numpy restatement of the reference's L1/L2 ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  All tensors are numpy arrays in the
reference's own layout: images/activations CHW (`helpers/utils.mojo:805-811`), token tensors
(T, D).  dtype defaults to float32 like the reference (`helpers/utils.mojo:12-15`); pass
float64 arrays for a higher-precision ground truth.

Citations are `file:line` under /root/reference.  "App.A Dn" = SURVEY.md Appendix A deviation.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Semantics:
    """Deterministic literal quirks of the reference kept as documentation flags (CPU only,
    never graded on GPU).  All False = the build-implemented semantics (App.A 'Build implements')."""
    literal_head_split: bool = False        # App.A D5  flat reinterpretation instead of view+transpose
    literal_softmax_axis: bool = False      # App.A D6  normalise over queries instead of keys
    literal_layernorm_global: bool = False  # App.A D8  one mean/sigma over all C*HW elements
    literal_time_freqs: bool = False        # App.A D9  (-i/160)**10000 instead of 10000**(-i/160)


DEFAULT = Semantics()

# ---------------------------------------------------------------------------------------------
# elementwise


def silu(x):
    """`SiLU.forward` helpers/utils.mojo:1892-1902: x / (1 + exp(-x)) (pure; App.A D16)."""
    return x / (1.0 + np.exp(-x))


def gelu_tanh(x):
    """`Gelu.forward` helpers/utils.mojo:1908-1919: tanh approximation (:1914)."""
    c = np.sqrt(np.asarray(2.0 / np.pi, dtype=x.dtype))
    return x * (0.5 * (1.0 + np.tanh(c * (x + x.dtype.type(0.044715) * x ** 3))))


def quick_gelu(x):
    """CLIP's activation, intended form x * sigmoid(1.702 x) (clip.mojo:49-50; literal aliasing bug App.A D15)."""
    return x / (1.0 + np.exp(-x * x.dtype.type(1.702)))


def embedding(tokens, table):
    """`Embedding.forward` helpers/utils.mojo:2032-2046, intended out[t] = W[token[t]] (App.A D3)."""
    return table[np.asarray(tokens, dtype=np.int64)]


def time_embedding(t, sem=DEFAULT, dtype=np.float32):
    """`get_time_embedding` helpers/utils.mojo:353-370 -> (320,).

    Build semantics (App.A D9): f_i = 10000**(-i/160), out = [cos(t f), sin(t f)].
    """
    i = np.arange(160, dtype=np.float64)
    if sem.literal_time_freqs:
        with np.errstate(under="ignore"):
            f = ((-i / 160.0) ** 10000).astype(np.float32).astype(np.float64)  # :361 literal
    else:
        f = 10000.0 ** (-i / 160.0)
    f = f.astype(dtype)
    x = f * dtype(t)
    return np.concatenate([np.cos(x), np.sin(x)]).astype(dtype)


# ---------------------------------------------------------------------------------------------
# pad / conv / upsample / concat


def pad(x, pad_h=(0, 0), pad_w=(0, 0)):
    """`Matrix.pad` helpers/utils.mojo:1383-1413: zero pad (top,bottom),(left,right) of CHW."""
    return np.pad(x, ((0, 0), tuple(pad_h), tuple(pad_w)))


def conv2d(x, w, b=None, padding=(0, 0), stride=(1, 1), pad_hw=None, max_cols_bytes=1 << 28):
    """`Conv2D.forward` helpers/utils.mojo:1738-1811.

    Cross-correlation with OIHW weights (:1718), symmetric zero padding (ph, pw) on both sides
    (:1744-1747), stride (sy, sx), Ho = floor((H+2p-k)/s)+1 (:1752-1758), + bias[o] (:1782).
    Reads only the first `in_channels` = w.shape[1] channels of x (:1771; App.A D11).
    `pad_hw=((t,b),(l,r))` overrides `padding` for the encoder's asymmetric pad (vae.mojo:115-116).
    """
    O, I, k, k2 = w.shape
    assert k == k2
    x = x[:I]
    if pad_hw is None:
        pad_hw = ((padding[0], padding[0]), (padding[1], padding[1]))
    xp = pad(x, pad_hw[0], pad_hw[1])
    C, H, W = xp.shape
    sy, sx = stride
    Ho = (H - k) // sy + 1
    Wo = (W - k) // sx + 1
    out = np.empty((O, Ho, Wo), dtype=x.dtype)
    wm = w.reshape(O, I * k * k)
    # windows: (C, Ho, Wo, k, k) view
    sC, sH, sW = xp.strides
    win = np.lib.stride_tricks.as_strided(
        xp, shape=(C, Ho, Wo, k, k), strides=(sC, sH * sy, sW * sx, sH, sW), writeable=False)
    rows_per = max(1, int(max_cols_bytes // max(1, I * k * k * Wo * x.dtype.itemsize)))
    for y0 in range(0, Ho, rows_per):
        y1 = min(Ho, y0 + rows_per)
        cols = win[:, y0:y1].transpose(0, 3, 4, 1, 2).reshape(I * k * k, (y1 - y0) * Wo)
        out[:, y0:y1] = (wm @ cols).reshape(O, y1 - y0, Wo)
    if b is not None:
        out += b.reshape(O, 1, 1).astype(x.dtype)
    return out


def upsample_nearest2x(x):
    """`Upsample.forward` helpers/utils.mojo:1979-2010 -> build semantics (App.A D1):
    nearest-neighbour x2 in H and W, channels unchanged, regardless of `scale_factor`."""
    return x.repeat(2, axis=1).repeat(2, axis=2)


def concat_channels(a, b):
    """`Matrix.concat(dim 0)` helpers/utils.mojo:605-647."""
    return np.concatenate([a, b], axis=0)


# ---------------------------------------------------------------------------------------------
# norms


def group_norm(x, num_groups, num_channels=None, eps=1e-5):
    """`GroupNorm.forward` helpers/utils.mojo:1845-1885 (App.A D12).

    Per group g over channels [g*Cn/G,(g+1)*Cn/G) x H x W: mu = mean, sigma = POPULATION std
    (:1372-1380), y = (x-mu)/(sigma+eps)*gamma with eps added to sigma, not inside sqrt
    (:1871-1873), scalar gamma = 1, beta unused (:1833-1834).  Only the first `num_channels`
    channels are normalised and returned (:1847,:1857-1859; App.A D11 - the rest is ignored
    by every consumer).
    """
    if num_channels is None:
        num_channels = x.shape[0]
    assert num_channels <= x.shape[0] and num_channels % num_groups == 0
    xs = x[:num_channels]
    C, H, W = xs.shape
    g = xs.reshape(num_groups, -1)
    mu = g.mean(axis=1, keepdims=True)
    sigma = np.sqrt(((g - mu) ** 2).mean(axis=1, keepdims=True))
    y = (g - mu) / (sigma + x.dtype.type(eps))
    return y.reshape(C, H, W)


def layer_norm(x, eps=1e-5, sem=DEFAULT):
    """`LayerNorm` helpers/utils.mojo:2052-2061 on a token tensor (T, C).

    Build semantics (App.A D8): per-token normalisation over C with the GroupNorm formula
    (x-mu)/(sigma+eps), population sigma, no affine.  literal_layernorm_global: one mu/sigma
    over all T*C elements (= GroupNorm(1, C) on the (C, T, 1) view, diffusion.mojo:120-123).
    """
    if sem.literal_layernorm_global:
        mu = x.mean()
        sigma = np.sqrt(((x - mu) ** 2).mean())
        return (x - mu) / (sigma + x.dtype.type(eps))
    mu = x.mean(axis=-1, keepdims=True)
    sigma = np.sqrt(((x - mu) ** 2).mean(axis=-1, keepdims=True))
    return (x - mu) / (sigma + x.dtype.type(eps))


# ---------------------------------------------------------------------------------------------
# linear / matmul / softmax / attention


def linear(x, w, b=None):
    """`Linear.forward` helpers/utils.mojo:1954-1976: y = x W^T (+ b); W (out,in) (:1943).
    Bias broadcast over rows (App.A D2)."""
    y = x @ w.T
    if b is not None:
        y = y + b.astype(x.dtype)
    return y


def matmul(a, b):
    """`Matrix.matmul` helpers/utils.mojo:1549-1569: batched over dim0, B broadcast when B.dim0==1."""
    return a @ b


def softmax_lastdim(s, sem=DEFAULT):
    """`Softmax` helpers/utils.mojo:411-448 as called by attention (dim=2, helpers/attention.mojo:59,112).

    Build semantics (App.A D6): softmax over the key (last) axis, with max-subtraction
    (mathematically identical; the literal code has none, :413).  literal_softmax_axis
    normalises each column, i.e. over queries (:435-445)."""
    axis = -2 if sem.literal_softmax_axis else -1
    e = np.exp(s - s.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _split_heads(x, H, sem):
    T, D = x.shape
    if sem.literal_head_split:
        return x.reshape(H, T, D // H)  # flat reinterpretation, helpers/utils.mojo:1134-1138
    return x.reshape(T, H, D // H).transpose(1, 0, 2)


def attention_core(q, k, v, H, causal=False, sem=DEFAULT):
    """softmax(q k^T / sqrt(d_h)) v with H heads; q (Tq,D), k/v (Tk,D) -> (Tq,D).
    helpers/attention.mojo:30-62 (self), :105-115 (cross).  Merge is the standard one (:61-62)."""
    Tq, D = q.shape
    dh = D // H
    qh, kh, vh = (_split_heads(t, H, sem) for t in (q, k, v))
    s = qh @ kh.transpose(0, 2, 1)
    if causal:  # intended mask j>i (App.A D7); only CLIP uses it (off-path)
        mask = np.triu(np.ones((Tq, k.shape[0]), dtype=bool), 1)
        s = np.where(mask, -np.inf, s)
    s = s / np.sqrt(np.asarray(dh, dtype=q.dtype))  # helpers/attention.mojo:57-58
    p = softmax_lastdim(s, sem)
    o = p @ vh  # (H, Tq, dh)
    return o.transpose(1, 0, 2).reshape(Tq, D)


def self_attention(x, H, w_in, b_in, w_out, b_out, causal=False, sem=DEFAULT):
    """`Self_Attention.forward` helpers/attention.mojo:26-65.
    qkv = x W_in^T (+b_in) -> chunk 3 on features (:29) -> heads -> attention -> out_proj (:63)."""
    qkv = linear(x, w_in, b_in)
    q, k, v = np.split(qkv, 3, axis=-1)
    o = attention_core(q, k, v, H, causal, sem)
    return linear(o, w_out, b_out)


def cross_attention(x, ctx, H, wq, bq, wk, bk, wv, bv, wo, bo, sem=DEFAULT):
    """`Cross_Attention.forward` helpers/attention.mojo:96-118."""
    q = linear(x, wq, bq)
    k = linear(ctx, wk, bk)
    v = linear(ctx, wv, bv)
    o = attention_core(q, k, v, H, False, sem)
    return linear(o, wo, bo)


# ---------------------------------------------------------------------------------------------
# layout helpers (CHW <-> tokens), diffusion.mojo:118-120,144-145 / vae.mojo:20-24 (App.A.3)


def chw_to_tokens(x):
    C, H, W = x.shape
    return x.reshape(C, H * W).T.copy()


def tokens_to_chw(t, H, W):
    return t.T.reshape(t.shape[1], H, W).copy()


def rescale_to_u8_range(x):
    """pipeline.mojo:127 `images.rescale((-1,1),(0,255),clamp=True)` (helpers/utils.mojo:577-597)."""
    return np.clip((x + 1.0) * 127.5, 0.0, 255.0).astype(x.dtype)


# ---------------------------------------------------------------------------------------------
# Extension (NOT reference behaviour): the norms PyTorch-trained checkpoints assume.
# Pinned against torch.nn.functional.group_norm / layer_norm in tests/test_oracle_pins.py.


def group_norm_torch(x, num_groups, eps=1e-5, weight=None, bias=None):
    """x (C,H,W): (x - mu) / sqrt(var + eps) * weight[c] + bias[c], population variance per group."""
    C, H, W = x.shape
    g = x.astype(np.float64).reshape(num_groups, -1)
    mu = g.mean(axis=1, keepdims=True)
    var = g.var(axis=1, keepdims=True)
    y = ((g - mu) / np.sqrt(var + eps)).reshape(C, H, W)
    if weight is not None:
        y = y * np.asarray(weight, np.float64).reshape(C, 1, 1)
    if bias is not None:
        y = y + np.asarray(bias, np.float64).reshape(C, 1, 1)
    return y.astype(np.float32)


def layer_norm_torch(x, eps=1e-5, weight=None, bias=None):
    """x (M,C): per-row (x - mu) / sqrt(var + eps) * weight + bias."""
    x64 = x.astype(np.float64)
    mu = x64.mean(axis=-1, keepdims=True)
    var = x64.var(axis=-1, keepdims=True)
    y = (x64 - mu) / np.sqrt(var + eps)
    if weight is not None:
        y = y * np.asarray(weight, np.float64)
    if bias is not None:
        y = y + np.asarray(bias, np.float64)
    return y.astype(np.float32)


def gelu_erf(x):
    """Exact GELU 0.5 x (1 + erf(x / sqrt 2)) = torch.nn.functional.gelu (extension; the reference's Gelu is the tanh form)."""
    x64 = np.asarray(x, np.float64)
    try:
        from scipy.special import erf
        e = erf(x64 / np.sqrt(2.0))
    except ImportError:
        from math import erf
        e = np.vectorize(erf)(x64 / np.sqrt(2.0))
    return (0.5 * x64 * (1.0 + e)).astype(np.float32)
