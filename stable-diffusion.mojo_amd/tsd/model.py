"""Device-resident models (`tsd_model`) and the denoise session (`tsd_session`)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f32, lib, ptr, vp

KINDS = {"diffusion": _lib.MODEL_DIFFUSION, "decoder": _lib.MODEL_DECODER, "encoder": _lib.MODEL_ENCODER,
         "clip": _lib.MODEL_CLIP, "diffusion_sd15": _lib.MODEL_DIFFUSION_SD15,
         "diffusion_sd15_torch": _lib.MODEL_DIFFUSION_SD15_TORCH, "clip_torch": _lib.MODEL_CLIP_TORCH,
         "decoder_torch": _lib.MODEL_DECODER_TORCH, "encoder_torch": _lib.MODEL_ENCODER_TORCH}


def param_specs(kind):
    """[(name, shape, used, init_bound)] in struct-field DFS order (SURVEY.md App.C).  No GPU needed."""
    k = KINDS[kind] if isinstance(kind, str) else kind
    n = lib().tsd_model_param_count(k)
    out = []
    name = C.create_string_buffer(128)
    shape = (C.c_int64 * 4)()
    ndim, used, bound = C.c_int(), C.c_int(), C.c_float()
    for i in range(n):
        check(lib().tsd_model_param_info(k, i, name, 128, shape, C.byref(ndim), C.byref(used), C.byref(bound)))
        out.append((name.value.decode(), tuple(int(shape[j]) for j in range(ndim.value)), bool(used.value), bound.value))
    return out


def flop_count(kind, L, T=77):
    """Algorithmic GFLOP of one forward per sample."""
    return lib().tsd_flop_count(KINDS[kind] if isinstance(kind, str) else kind, L, T)


class Model:
    """Packed device weights of Diffusion / Decoder / Encoder."""

    def __init__(self, kind, ctx=None, seed=None):
        self.kind = KINDS[kind] if isinstance(kind, str) else kind
        self.ctx = ctx or _lib.default_context()
        h = vp()
        check(lib().tsd_model_create(self.ctx.h, self.kind, C.byref(h)))
        self.h = h
        self.specs = param_specs(self.kind)
        if seed is not None:
            self.init_random(seed)

    def init_random(self, seed):
        check(lib().tsd_model_init_random(self.h, int(seed)))

    def set_param(self, index, array):
        a = f32(array)
        check(lib().tsd_model_set_param(self.h, int(index), ptr(a), a.size))

    def load_params(self, params):
        """params: {name: array} in the reference layouts (conv OIHW, linear (out,in))."""
        for i, (name, shape, used, _) in enumerate(self.specs):
            if name in params:
                self.set_param(i, np.asarray(params[name]).reshape(shape))
            elif used:
                raise KeyError(f"missing parameter {name}")

    def packed_blob(self):
        p, n = vp(), C.c_size_t()
        check(lib().tsd_model_packed_blob(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def mark_loaded(self):
        check(lib().tsd_model_mark_loaded(self.h))

    def prepare(self):
        """Build the derived device buffers now (per rank, after the weights are in place) and wait for them."""
        check(lib().tsd_model_prepare(self.h))

    def close(self):
        if self.h:
            lib().tsd_model_destroy(self.h)
            self.h = None


class Session:
    """Device-resident denoise loop (pipeline.mojo:57-127 + sampler.mojo)."""

    def __init__(self, diffusion, decoder, B, L, T=77, cfg=False):
        h = vp()
        check(lib().tsd_session_create(diffusion.h, decoder.h if decoder is not None else None, B, L, T, 1 if cfg else 0,
                                       C.byref(h)))
        self.h, self.B, self.L, self.T, self.cfg = h, B, L, T, cfg
        self.ctx = diffusion.ctx
        self.has_decoder = decoder is not None

    def set_schedule(self, num_training_steps=1000, num_inference_steps=50, start_step=0):
        check(lib().tsd_session_set_schedule(self.h, num_training_steps, num_inference_steps, start_step))

    @property
    def num_steps(self):
        return lib().tsd_session_num_steps(self.h)

    def timestep(self, i):
        return lib().tsd_session_timestep(self.h, i)

    def upload(self, latents, context, uncond_context=None, noise=None, cfg_scale=7.5):
        la, cx = f32(latents), f32(context)
        uc = f32(uncond_context) if uncond_context is not None else None
        nz = f32(noise) if noise is not None else None
        # the C side reads exactly B*4*L*L / B*T*768 / steps*B*4*L*L floats from these pointers: check the shapes here
        B, L, T = self.B, self.L, self.T
        if la.shape != (B, 4, L, L):
            raise ValueError(f"latents must have shape {(B, 4, L, L)}, got {la.shape}")
        if cx.shape != (B, T, 768):
            raise ValueError(f"context must have shape {(B, T, 768)}, got {cx.shape}")
        if self.cfg and uc is None:
            raise ValueError("a CFG session needs uncond_context")
        if uc is not None:
            if uc.ndim == 2:
                uc = uc[None]
            if uc.shape == (1, T, 768) and B > 1:  # one shared negative prompt: broadcast it like Diffusion.forward does
                uc = np.ascontiguousarray(np.broadcast_to(uc, (B, T, 768)))
            if uc.shape != (B, T, 768):
                raise ValueError(f"uncond_context must have shape {(B, T, 768)} (or one shared (T, 768) row), got {uc.shape}")
        if nz is not None and nz.shape != (self.num_steps, B, 4, L, L):
            raise ValueError(f"noise must have shape {(self.num_steps, B, 4, L, L)}, got {nz.shape}")
        check(lib().tsd_session_upload(self.h, ptr(la), ptr(cx), ptr(uc), ptr(nz), float(cfg_scale)))

    def step(self, i):
        check(lib().tsd_session_step(self.h, int(i)))

    def add_noise(self, i, noise):
        check(lib().tsd_session_add_noise(self.h, int(i), ptr(f32(noise))))

    def decode(self):
        check(lib().tsd_session_decode(self.h))

    def latents(self):
        out = np.empty((self.B, 4, self.L, self.L), dtype=np.float32)
        check(lib().tsd_session_download_latents(self.h, ptr(out)))
        return out

    def images(self, rescale=True):
        out = np.empty((self.B, 3, 8 * self.L, 8 * self.L), dtype=np.float32)
        check(lib().tsd_session_download_images(self.h, 1 if rescale else 0, ptr(out)))
        return out

    def close(self):
        if self.h:
            lib().tsd_session_destroy(self.h)
            self.h = None
