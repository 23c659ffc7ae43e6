"""Extensions that are NOT reference behaviour: the norm semantics real, PyTorch-trained
checkpoints assume.  The reference's own `GroupNorm` / `LayerNorm` stay in tsd.utils."""
import numpy as np

from ._lib import NULL_MATRIX, check, f32, lib, ptr
from .utils import _ctx


class TorchGroupNorm:
    """torch.nn.GroupNorm(num_groups, num_channels, eps, affine): (x - mu) / sqrt(var + eps) * weight[c] + bias[c],
    optionally followed by a fused SiLU.  x is (C, H, W)."""

    def __init__(self, num_groups, num_channels, eps=1e-5, weight=None, bias=None, silu=False, ctx=None):
        self.num_groups, self.num_channels, self.eps, self.silu, self.ctx = num_groups, num_channels, eps, silu, ctx
        self.weight = None if weight is None else f32(weight).reshape(num_channels)
        self.bias = None if bias is None else f32(bias).reshape(num_channels)

    def forward(self, x):
        x = f32(x)
        C, H, W = x.shape
        y = np.empty_like(x)
        code = lib().tsd_groupnorm_affine_f32(_ctx(self.ctx), ptr(x), C, H, W, self.num_groups, self.eps,
                                              ptr(self.weight), ptr(self.bias), 1 if self.silu else 0, ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class TorchLayerNorm:
    """torch.nn.LayerNorm(n_embed, eps, elementwise_affine) over the last dim of x (M, C)."""

    def __init__(self, n_embed, eps=1e-5, weight=None, bias=None, ctx=None):
        self.n_embed, self.eps, self.ctx = n_embed, eps, ctx
        self.weight = None if weight is None else f32(weight).reshape(n_embed)
        self.bias = None if bias is None else f32(bias).reshape(n_embed)

    def forward(self, x):
        x = f32(x)
        M, C = x.shape
        y = np.empty_like(x)
        code = lib().tsd_layernorm_affine_f32(_ctx(self.ctx), ptr(x), M, C, self.eps, ptr(self.weight), ptr(self.bias),
                                              ptr(y))
        return NULL_MATRIX() if check(code, True) else y
