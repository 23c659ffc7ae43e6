"""Host-side mirror of the reference's `helpers/utils.mojo` layer structs (L1 ops).

Same struct names, constructor arguments and `forward()` meaning as the reference; bodies are one
call into libtsd.so (HIP kernels on gfx950).  Tensors are numpy float32 arrays in the reference's
layout: (C,H,W) images, (1,T,D) or (T,D) token matrices.  Error behaviour mirrors the reference:
shape errors print a message and return a null matrix (`Matrix(0,0,0)`).
"""
import math

import numpy as np

from . import _lib, rng
from ._lib import NULL_MATRIX, check, f32, lib, ptr


def _ctx(ctx):
    return (ctx or _lib.default_context()).h


def _tokens(x):
    """(1,T,D) or (T,D) -> (T,D), remembering whether to restore the leading 1."""
    x = f32(x)
    if x.ndim == 3:
        if x.shape[0] != 1:
            raise ValueError("token tensors are (1,T,D) in the reference")
        return x[0], True
    return x, False


class Conv2D:
    """`Conv2D` helpers/utils.mojo:1693-1811: OIHW kernel U(+-1/sqrt(cin*k*k)) (:1722-1724), zero bias (:1717)."""

    def __init__(self, in_channels, out_channels, kernel_size, padding=(0, 0), stride=(1, 1), seed=0, ctx=None):
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.padding, self.stride, self.ctx = tuple(padding), tuple(stride), ctx
        k = in_channels * kernel_size * kernel_size
        self.kernel = rng.uniform(seed, rng.fresh_id(), out_channels * k, 1.0 / math.sqrt(k)).reshape(
            out_channels, in_channels, kernel_size, kernel_size)
        self.bias = np.zeros(out_channels, dtype=np.float32)

    def forward(self, matrix):
        x = f32(matrix)
        C, H, W = x.shape
        ph, pw = self.padding
        sy, sx = self.stride
        k = self.kernel_size
        Ho, Wo = (H + 2 * ph - k) // sy + 1, (W + 2 * pw - k) // sx + 1
        y = np.empty((self.out_channels, max(Ho, 0), max(Wo, 0)), dtype=np.float32)
        code = lib().tsd_conv2d_f32(_ctx(self.ctx), ptr(x), C, H, W, ptr(f32(self.kernel)), ptr(f32(self.bias)),
                                    self.in_channels, self.out_channels, k, ph, pw, sy, sx, ptr(y))
        return NULL_MATRIX() if check(code, True) else y


def pad(matrix, pad_h=(0, 0), pad_w=(0, 0), ctx=None):
    """`Matrix.pad` helpers/utils.mojo:1383-1413."""
    x = f32(matrix)
    C, H, W = x.shape
    y = np.empty((C, H + sum(pad_h), W + sum(pad_w)), dtype=np.float32)
    code = lib().tsd_pad_f32(_ctx(ctx), ptr(x), C, H, W, pad_h[0], pad_h[1], pad_w[0], pad_w[1], ptr(y))
    return NULL_MATRIX() if check(code, True) else y


class GroupNorm:
    """`GroupNorm` helpers/utils.mojo:1813-1885: scalar gamma=1, beta unused, eps added to sigma."""

    def __init__(self, num_groups, num_channels, epsilon=1e-5, ctx=None):
        self.num_groups, self.num_channels, self.epsilon = num_groups, num_channels, epsilon
        self.gamma, self.beta, self.ctx = 1.0, 0.0, ctx

    def forward(self, x):
        x = f32(x)
        C, H, W = x.shape
        y = np.empty((self.num_channels, H, W), dtype=np.float32)
        code = lib().tsd_groupnorm_f32(_ctx(self.ctx), ptr(x), C, H, W, self.num_groups, self.num_channels,
                                       self.epsilon, self.gamma, ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class LayerNorm:
    """`LayerNorm` helpers/utils.mojo:2052-2061 with the build semantics (per-token, SURVEY.md App.A D8)."""

    def __init__(self, n_embed, ctx=None):
        self.n_embed, self.ctx = n_embed, ctx

    def forward(self, x):
        t, lead = _tokens(x)
        y = np.empty_like(t)
        code = lib().tsd_layernorm_f32(_ctx(self.ctx), ptr(t), t.shape[0], t.shape[1], 1e-5, ptr(y))
        if check(code, True):
            return NULL_MATRIX()
        return y[None] if lead else y


def _unary(fn_name, x, ctx):
    x = f32(x)
    y = np.empty_like(x)
    check(getattr(lib(), fn_name)(_ctx(ctx), ptr(x), x.size, ptr(y)))
    return y


class SiLU:
    """`SiLU` helpers/utils.mojo:1888-1902 (pure - App.A D16)."""

    def __init__(self, ctx=None):
        self.ctx = ctx

    def forward(self, x):
        return _unary("tsd_silu_f32", x, self.ctx)


class Gelu:
    """`Gelu` helpers/utils.mojo:1904-1919 (tanh approximation)."""

    def __init__(self, ctx=None):
        self.ctx = ctx

    def forward(self, x):
        return _unary("tsd_gelu_tanh_f32", x, self.ctx)


class Linear:
    """`Linear` helpers/utils.mojo:1921-1976: weight (out,in), bias (out) allocated always, used iff use_bias.
    Synthetic init U(+-1/sqrt(in)) (App.A D18)."""

    def __init__(self, in_features, out_features, use_bias=True, seed=0, ctx=None):
        self.in_features, self.out_features, self.use_bias, self.ctx = in_features, out_features, use_bias, ctx
        b = 1.0 / math.sqrt(in_features)
        self.weight = rng.uniform(seed, rng.fresh_id(), out_features * in_features, b).reshape(out_features, in_features)
        self.bias = rng.uniform(seed, rng.fresh_id(), out_features, b)

    def forward(self, x):
        t, lead = _tokens(x)
        if t.shape[1] != self.in_features:  # helpers/utils.mojo:1955-1957
            print("Invalid input dimensions for Linear layer. Returning null matrix")
            return NULL_MATRIX()
        y = np.empty((t.shape[0], self.out_features), dtype=np.float32)
        code = lib().tsd_linear_f32(_ctx(self.ctx), ptr(t), t.shape[0], self.in_features, ptr(f32(self.weight)),
                                    ptr(f32(self.bias)) if self.use_bias else None, self.out_features, ptr(y))
        if check(code, True):
            return NULL_MATRIX()
        return y[None] if lead else y


def matmul(a, b, ctx=None):
    """`Matrix.matmul` helpers/utils.mojo:1549-1569: (Ba,M,K) x (Bb,K,N), Bb == 1 broadcasts."""
    a, b = f32(a), f32(b)
    if a.shape[2] != b.shape[1]:  # :1550-1552
        print("Incompatible dimensions for matrix multiplication. Returning null matrix")
        return NULL_MATRIX()
    y = np.empty((a.shape[0], a.shape[1], b.shape[2]), dtype=np.float32)
    code = lib().tsd_matmul_f32(_ctx(ctx), ptr(a), ptr(b), a.shape[0], b.shape[0], a.shape[1], a.shape[2], b.shape[2], ptr(y))
    return NULL_MATRIX() if check(code, True) else y


class Upsample:
    """`Upsample` helpers/utils.mojo:1979-2010, build semantics (App.A D1): nearest x2 whatever `scale_factor`."""

    def __init__(self, scale_factor=1, ctx=None):
        self.scale_factor, self.ctx = scale_factor, ctx

    def forward(self, x):
        if self.scale_factor < 1:  # :1991-1993
            print("Invalid scale factor for upsampling. Returning null matrix")
            return NULL_MATRIX()
        x = f32(x)
        C, H, W = x.shape
        y = np.empty((C, 2 * H, 2 * W), dtype=np.float32)
        check(lib().tsd_upsample_nearest2x_f32(_ctx(self.ctx), ptr(x), C, H, W, ptr(y)))
        return y


def Softmax(matrix, dim=2, ctx=None):
    """`Softmax` helpers/utils.mojo:411-448 as attention uses it; build semantics (App.A D6): last axis."""
    if dim != 2:
        print("Invalid dimension for softmax. Returning null matrix")  # only the attention axis is on the path
        return NULL_MATRIX()
    x = f32(matrix)
    y = np.empty_like(x)
    check(lib().tsd_softmax_lastdim_f32(_ctx(ctx), ptr(x), int(np.prod(x.shape[:-1])), x.shape[-1], ptr(y)))
    return y


def get_time_embedding(timestep, ctx=None):
    """`get_time_embedding` helpers/utils.mojo:353-370 -> (1,1,320)."""
    y = np.empty(320, dtype=np.float32)
    check(lib().tsd_time_embedding_f32(_ctx(ctx), float(timestep), ptr(y)))
    return y.reshape(1, 1, 320)


def concat(a, b, dim=0):
    """`Matrix.concat` helpers/utils.mojo:605-647 (host glue, like the reference's Matrix method)."""
    return np.concatenate([f32(a), f32(b)], axis=dim)


def rescale(x, old_scale, new_scale, clamp=False):
    """`Matrix.rescale` helpers/utils.mojo:577-597 (host glue; pipeline.mojo:70,127)."""
    o0, o1 = old_scale
    n0, n1 = new_scale
    y = (f32(x) - o0) * (n1 - n0) / (o1 - o0) + n0
    return np.clip(y, n0, n1).astype(np.float32) if clamp else y.astype(np.float32)
