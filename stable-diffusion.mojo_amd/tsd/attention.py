"""Host-side mirror of `helpers/attention.mojo`: Self_Attention / Cross_Attention (one libtsd call each)."""
import numpy as np

from . import _lib
from ._lib import NULL_MATRIX, check, f32, lib, ptr
from .utils import Linear, _ctx, _tokens


class Self_Attention:
    """`Self_Attention` helpers/attention.mojo:5-65."""

    def __init__(self, n_heads, d_embedding, in_bias=True, out_bias=True, seed=0, ctx=None):
        self.n_heads, self.ctx = n_heads, ctx
        self.in_proj = Linear(d_embedding, 3 * d_embedding, in_bias, seed=seed, ctx=ctx)
        self.out_proj = Linear(d_embedding, d_embedding, out_bias, seed=seed, ctx=ctx)

    def forward(self, x, causal_mask=False):
        t, lead = _tokens(x)
        T, D = t.shape
        y = np.empty((T, D), dtype=np.float32)
        ip, op = self.in_proj, self.out_proj
        code = lib().tsd_self_attention_f32(
            _ctx(self.ctx), ptr(t), T, D, self.n_heads, ptr(f32(ip.weight)), ptr(f32(ip.bias)) if ip.use_bias else None,
            ptr(f32(op.weight)), ptr(f32(op.bias)) if op.use_bias else None, 1 if causal_mask else 0, ptr(y))
        if check(code, True):
            return NULL_MATRIX()
        return y[None] if lead else y


class Cross_Attention:
    """`Cross_Attention` helpers/attention.mojo:68-118."""

    def __init__(self, n_heads, d_embedding, d_crossing, in_bias=True, out_bias=True, seed=0, ctx=None):
        self.n_heads, self.ctx = n_heads, ctx
        self.q_proj = Linear(d_embedding, d_embedding, in_bias, seed=seed, ctx=ctx)
        self.k_proj = Linear(d_crossing, d_embedding, in_bias, seed=seed, ctx=ctx)
        self.v_proj = Linear(d_crossing, d_embedding, in_bias, seed=seed, ctx=ctx)
        self.out_proj = Linear(d_embedding, d_embedding, out_bias, seed=seed, ctx=ctx)

    def forward(self, x, context):
        t, lead = _tokens(x)
        c, _ = _tokens(context)
        Tq, D = t.shape
        Tk, Dc = c.shape
        y = np.empty((Tq, D), dtype=np.float32)

        def wb(l):
            return ptr(f32(l.weight)), (ptr(f32(l.bias)) if l.use_bias else None)

        (wq, bq), (wk, bk), (wv, bv), (wo, bo) = wb(self.q_proj), wb(self.k_proj), wb(self.v_proj), wb(self.out_proj)
        code = lib().tsd_cross_attention_f32(_ctx(self.ctx), ptr(t), Tq, D, ptr(c), Tk, Dc, self.n_heads,
                                             wq, bq, wk, bk, wv, bv, wo, bo, ptr(y))
        if check(code, True):
            return NULL_MATRIX()
        return y[None] if lead else y
