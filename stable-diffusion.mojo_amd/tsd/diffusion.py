"""Host-side mirror of `diffusion.mojo`: same struct names, constructor arguments and forward() meaning.

`Diffusion` is the measured module (device-resident weights, one C call per forward).  The block
structs (`Time_Embedding`, `Unet_Residual_Block`, `Unet_Attention_Block`, `UNet_Output_Layer`) are
one C call per block with host weights, and `UNet` composes them exactly like diffusion.mojo:228-273 -
the per-struct drop-in surface.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import NULL_MATRIX, check, f32, fp, lib, ptr
from .attention import Cross_Attention, Self_Attention
from .model import Model
from .utils import Conv2D, GroupNorm, LayerNorm, Linear, SiLU, Upsample, _ctx, concat


class Time_Embedding:
    """`Time_Embedding` diffusion.mojo:5-21."""

    def __init__(self, n_embed, seed=0, ctx=None):
        self.ctx = ctx
        self.layer1 = Linear(n_embed, 4 * n_embed, seed=seed, ctx=ctx)
        self.layer2 = Linear(4 * n_embed, 4 * n_embed, seed=seed, ctx=ctx)

    def forward(self, x):
        t = f32(x).reshape(-1)
        if t.size != 320 or self.layer1.out_features != 1280:
            print("Invalid input dimensions for Time_Embedding. Returning null matrix")
            return NULL_MATRIX()
        y = np.empty(1280, dtype=np.float32)
        check(lib().tsd_time_embedding_mlp_f32(_ctx(self.ctx), ptr(t), ptr(f32(self.layer1.weight)), ptr(f32(self.layer1.bias)),
                                               ptr(f32(self.layer2.weight)), ptr(f32(self.layer2.bias)), ptr(y)))
        return y.reshape(1, 1, 1280)


class Unet_Residual_Block:
    """`Unet_Residual_Block` diffusion.mojo:24-72."""

    def __init__(self, in_channels, out_channels, n_time=1280, seed=0, ctx=None):
        self.in_channels, self.out_channels, self.ctx = in_channels, out_channels, ctx
        self.layer1 = GroupNorm(32, in_channels)
        self.layer2 = Conv2D(in_channels, out_channels, 3, (1, 1), seed=seed)
        self.layer3 = Linear(n_time, out_channels, seed=seed)
        self.layer4 = GroupNorm(32, out_channels)
        self.layer5 = Conv2D(out_channels, out_channels, 3, (1, 1), seed=seed)
        self.layer6 = Conv2D(in_channels, out_channels, 1, (0, 0), seed=seed)

    def forward(self, x, time):
        x = f32(x)
        Cx, H, W = x.shape
        y = np.empty((self.out_channels, H, W), dtype=np.float32)
        t = f32(time).reshape(-1)
        code = lib().tsd_unet_residual_block_f32(
            _ctx(self.ctx), ptr(x), Cx, H, W, ptr(t), self.in_channels, self.out_channels,
            ptr(f32(self.layer2.kernel)), ptr(f32(self.layer2.bias)), ptr(f32(self.layer3.weight)), ptr(f32(self.layer3.bias)),
            ptr(f32(self.layer5.kernel)), ptr(f32(self.layer5.bias)), ptr(f32(self.layer6.kernel)), ptr(f32(self.layer6.bias)),
            ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class Unet_Attention_Block:
    """`Unet_Attention_Block` diffusion.mojo:75-147."""

    def __init__(self, n_head, n_embed, d_context=768, seed=0, ctx=None):
        channels = n_head * n_embed
        self.n_head, self.n_embed, self.ctx = n_head, n_embed, ctx
        self.layer1 = GroupNorm(32, channels, epsilon=1e-6)
        self.layer2 = Conv2D(channels, channels, 1, (0, 0), seed=seed)
        self.layer3 = LayerNorm(channels)
        self.layer4 = Self_Attention(n_head, channels, in_bias=False, seed=seed)
        self.layer5 = LayerNorm(channels)
        self.layer6 = Cross_Attention(n_head, channels, d_context, in_bias=False, seed=seed)
        self.layer7 = LayerNorm(channels)
        self.layer8 = Linear(channels, 8 * channels, seed=seed)
        self.layer9 = Linear(4 * channels, channels, seed=seed)
        self.layer10 = Conv2D(channels, channels, 1, (0, 0), seed=seed)

    def weight_list(self):
        l4, l6 = self.layer4, self.layer6
        return [self.layer2.kernel, self.layer2.bias, l4.in_proj.weight, l4.out_proj.weight, l4.out_proj.bias,
                l6.q_proj.weight, l6.k_proj.weight, l6.v_proj.weight, l6.out_proj.weight, l6.out_proj.bias,
                self.layer8.weight, self.layer8.bias, self.layer9.weight, self.layer9.bias,
                self.layer10.kernel, self.layer10.bias]

    def forward(self, x, context):
        x = f32(x)
        Cx, H, W = x.shape
        c = f32(context)
        c = c[0] if c.ndim == 3 else c
        ws = [f32(w) for w in self.weight_list()]
        arr = (fp * len(ws))(*[ptr(w) for w in ws])
        y = np.empty_like(x)
        code = lib().tsd_unet_attention_block_f32(_ctx(self.ctx), ptr(x), self.n_head, self.n_embed, H, W, ptr(c),
                                                  c.shape[0], c.shape[1], arr, len(ws), ptr(y))
        return NULL_MATRIX() if check(code, True) else y


class UNet:
    """`UNet` diffusion.mojo:150-273 composed from the block structs (per-struct drop-in path)."""

    def __init__(self, seed=0, ctx=None):
        R, A = Unet_Residual_Block, Unet_Attention_Block
        k = dict(seed=seed, ctx=ctx)
        self.layer1 = Conv2D(4, 320, 3, (1, 1), **k)
        self.layer2, self.layer3 = R(320, 320, **k), A(8, 40, **k)
        self.layer4 = Conv2D(320, 320, 3, (1, 1), (2, 2), **k)
        self.layer5, self.layer6 = R(320, 640, **k), A(8, 80, **k)
        self.layer7 = Conv2D(640, 640, 3, (1, 1), (2, 2), **k)
        self.layer8, self.layer9 = R(640, 1280, **k), A(8, 160, **k)
        self.layer10, self.layer11 = R(2560, 1280, **k), A(8, 160, **k)
        self.layer12, self.layer13 = R(1920, 1280, **k), A(8, 160, **k)
        self.layer14 = Upsample(1280, ctx=ctx)
        self.layer15, self.layer16 = R(1280, 640, **k), A(8, 80, **k)
        self.layer17, self.layer18 = R(960, 640, **k), A(8, 80, **k)
        self.layer19 = Upsample(640, ctx=ctx)
        self.layer20, self.layer21 = R(640, 320, **k), A(8, 40, **k)
        self.layer22, self.layer23 = R(640, 320, **k), A(8, 40, **k)

    def forward(self, x, context, time):
        out = self.layer1.forward(x); skip1 = out
        out = self.layer2.forward(out, time); out = self.layer3.forward(out, context); skip2 = out
        out = self.layer4.forward(out); skip3 = out
        out = self.layer5.forward(out, time); out = self.layer6.forward(out, context); skip4 = out
        out = self.layer7.forward(out); skip5 = out
        out = self.layer8.forward(out, time); out = self.layer9.forward(out, context); skip6 = out
        out = concat(out, skip6, 0); out = self.layer10.forward(out, time); out = self.layer11.forward(out, context)
        out = concat(out, skip5, 0); out = self.layer12.forward(out, time); out = self.layer13.forward(out, context)
        out = self.layer14.forward(out)
        out = concat(out, skip4, 0); out = self.layer15.forward(out, time); out = self.layer16.forward(out, context)
        out = concat(out, skip3, 0); out = self.layer17.forward(out, time); out = self.layer18.forward(out, context)
        out = self.layer19.forward(out)
        out = concat(out, skip2, 0); out = self.layer20.forward(out, time); out = self.layer21.forward(out, context)
        out = concat(out, skip1, 0); out = self.layer22.forward(out, time); out = self.layer23.forward(out, context)
        return out


class UNet_Output_Layer:
    """`UNet_Output_Layer` diffusion.mojo:275-291 (composed: GroupNorm(320 groups) -> SiLU -> Conv2D)."""

    def __init__(self, in_channels, out_channels, seed=0, ctx=None):
        self.layer1 = GroupNorm(320, in_channels, ctx=ctx)
        self.layer2 = Conv2D(in_channels, out_channels, 3, (1, 1), seed=seed, ctx=ctx)
        self.ctx = ctx

    def forward(self, x):
        out = self.layer1.forward(x)
        out = SiLU(self.ctx).forward(out)
        return self.layer2.forward(out)


class Diffusion:
    """`Diffusion` diffusion.mojo:294-318 - device-resident weights; forward is ONE libtsd call.

    forward(x, context, time): x (4,L,L) or (B,4,L,L); context (T,768)/(1,T,768) or (B,T,768);
    time = get_time_embedding(t): (1,1,320) or (B,320)."""

    def __init__(self, seed=0, ctx=None, params=None, variant="diffusion"):
        """variant: "diffusion" (the reference's 23-layer Tiny-SD graph) or "diffusion_sd15" (full-size 860 M UNet,
        BASELINE configs[4]; same blocks, not defined by the reference) or "diffusion_sd15_torch" (that graph with the
        norm semantics and parameters of PyTorch-trained SD-1.x checkpoints, see tsd.checkpoint)."""
        self.model = Model(variant, ctx=ctx, seed=None if params is not None else seed)
        if params is not None:
            self.model.load_params(params)

    def forward(self, x, context, time):
        x = f32(x)
        single = x.ndim == 3
        xb = x[None] if single else x
        B, _, L, _ = xb.shape
        c = f32(context)
        if c.ndim == 2:
            c = c[None]
        if c.shape[0] == 1 and B > 1:
            c = np.repeat(c, B, axis=0)
        t = f32(time).reshape(-1, 320)
        if t.shape[0] == 1 and B > 1:
            t = np.repeat(t, B, axis=0)
        c, t = f32(c), f32(t)
        out = np.empty_like(xb)
        code = lib().tsd_diffusion_forward(self.model.h, ptr(xb), ptr(c), ptr(t), B, L, c.shape[1], ptr(out))
        if check(code, True):
            return NULL_MATRIX()
        return out[0] if single else out
