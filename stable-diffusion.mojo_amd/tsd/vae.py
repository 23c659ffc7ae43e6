"""Host-side mirror of `vae.mojo`: Attention_Block, Res_Block (one libtsd call each), Decoder / Encoder
(device-resident weights, one call per forward)."""
import numpy as np

from ._lib import NULL_MATRIX, check, f32, lib, ptr
from .attention import Self_Attention
from .model import Model
from .utils import Conv2D, GroupNorm, _ctx


class Attention_Block:
    """VAE `Attention_Block` vae.mojo:5-27."""

    def __init__(self, channels, seed=0, ctx=None):
        self.ctx = ctx
        self.group_norm = GroupNorm(32, channels)
        self.attention = Self_Attention(1, channels, seed=seed)

    def forward(self, x):
        x = f32(x)
        C, H, W = x.shape
        y = np.empty_like(x)
        a = self.attention
        code = lib().tsd_vae_attention_block_f32(_ctx(self.ctx), ptr(x), C, H, W, ptr(f32(a.in_proj.weight)),
                                                 ptr(f32(a.in_proj.bias)), ptr(f32(a.out_proj.weight)),
                                                 ptr(f32(a.out_proj.bias)), ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class Res_Block:
    """VAE `Res_Block` vae.mojo:30-67 (GroupNorm with 16 groups)."""

    def __init__(self, in_channels, out_channels, seed=0, ctx=None):
        self.in_channels, self.out_channels, self.ctx = in_channels, out_channels, ctx
        self.group_norm1, self.group_norm2 = GroupNorm(16, in_channels), GroupNorm(16, out_channels)
        self.conv1 = Conv2D(in_channels, out_channels, 3, (1, 1), seed=seed)
        self.conv2 = Conv2D(out_channels, out_channels, 3, (1, 1), seed=seed)
        self.res_conv_layer = Conv2D(in_channels, out_channels, 1, seed=seed)

    def forward(self, x):
        x = f32(x)
        C, H, W = x.shape
        if C != self.in_channels:
            print("Invalid input dimensions for Res_Block. Returning null matrix")
            return NULL_MATRIX()
        y = np.empty((self.out_channels, H, W), dtype=np.float32)
        code = lib().tsd_vae_res_block_f32(
            _ctx(self.ctx), ptr(x), H, W, self.in_channels, self.out_channels, ptr(f32(self.conv1.kernel)),
            ptr(f32(self.conv1.bias)), ptr(f32(self.conv2.kernel)), ptr(f32(self.conv2.bias)),
            ptr(f32(self.res_conv_layer.kernel)), ptr(f32(self.res_conv_layer.bias)), ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class Decoder:
    """`Decoder` vae.mojo:162-250.  forward(x): (4,L,L) or (B,4,L,L) -> (3,8L,8L) / (B,3,8L,8L)."""

    def __init__(self, seed=0, ctx=None, params=None, variant="decoder"):
        """variant "decoder_torch" (extension): the trained VAE's norms (32 groups, per-channel affine) for real
        checkpoints (tsd.checkpoint.load_vae)."""
        self.model = Model(variant, ctx=ctx, seed=None if params is not None else seed)
        if params is not None:
            self.model.load_params(params)

    def forward(self, x):
        x = f32(x)
        single = x.ndim == 3
        xb = x[None] if single else x
        B, _, L, _ = xb.shape
        out = np.empty((B, 3, 8 * L, 8 * L), dtype=np.float32)
        code = lib().tsd_decoder_forward(self.model.h, ptr(xb), B, L, ptr(out))
        if check(code, True):
            return NULL_MATRIX()
        return out[0] if single else out


class Encoder:
    """`Encoder` vae.mojo:70-159.  forward(x, noise): (3,S,S),(4,S/8,S/8) -> (4,S/8,S/8) (batched variants too)."""

    def __init__(self, seed=0, ctx=None, params=None, variant="encoder"):
        self.model = Model(variant, ctx=ctx, seed=None if params is not None else seed)
        if params is not None:
            self.model.load_params(params)

    def forward(self, x, noise):
        x, noise = f32(x), f32(noise)
        single = x.ndim == 3
        xb = x[None] if single else x
        nb = noise[None] if noise.ndim == 3 else noise
        B, _, S, _ = xb.shape
        out = np.empty((B, 4, S // 8, S // 8), dtype=np.float32)
        code = lib().tsd_encoder_forward(self.model.h, ptr(xb), ptr(f32(nb)), B, S, ptr(out))
        if check(code, True):
            return NULL_MATRIX()
        return out[0] if single else out
