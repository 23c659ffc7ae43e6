"""Host-side mirror of `clip.mojo`: CLIP text encoder (the step before the hot path).

`CLIP.forward(tokens)` is one call through the C ABI (`tsd_clip_forward`) with device-resident packed weights;
intended semantics only."""
import ctypes as C

import numpy as np

from ._lib import NULL_MATRIX, check, lib
from .model import Model


class CLIP:
    """`CLIP` clip.mojo:56-109: ClipEmbedding(49408, 768, 77) + 12 x ClipPlayer(12, 768) + LayerNorm(768).

    forward(tokens): int ids, shape (T,) or (B, T) with T <= 77 (zero-padded to 77 like clip.mojo:91-93)
    -> (77, 768) or (B, 77, 768) float32: the `context` of Diffusion.forward / generate."""

    def __init__(self, seed=0, ctx=None, params=None, variant="clip"):
        """variant "clip_torch" (extension): torch LayerNorms with weight / bias = Hugging Face's CLIPTextModel, for real
        checkpoints (tsd.checkpoint.load_clip_text)."""
        self.model = Model(variant, ctx=ctx, seed=None if params is not None else seed)
        if params is not None:
            self.model.load_params(params)

    def forward(self, tokens):
        t = np.ascontiguousarray(np.asarray(tokens), dtype=np.int32)
        single = t.ndim == 1
        tb = t[None] if single else t
        B, T = tb.shape
        out = np.empty((B, 77, 768), dtype=np.float32)
        code = lib().tsd_clip_forward(self.model.h, tb.ctypes.data_as(C.POINTER(C.c_int32)), B, T,
                                      out.ctypes.data_as(C.POINTER(C.c_float)))
        if check(code, True):
            return NULL_MATRIX()
        return out[0] if single else out
