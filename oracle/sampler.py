"""DDPM sampler + denoise-loop oracle (App.A.2 'DDPM').

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows `sampler.mojo:15-124` and the loop of
`pipeline.mojo:57-127`.  Noise is an INPUT (App.A D19): callers pass z ~ N(0,1) per step.
"""
import numpy as np

from . import models, ops
from .ops import DEFAULT


class DDPMSampler:
    """`DDPMSampler` sampler.mojo:5-124 with num_training_steps a parameter (App.A D22)."""

    def __init__(self, num_training_steps=1000, beta_start=0.00085, beta_end=0.0120):
        self.num_training_steps = num_training_steps
        # :28-30  betas = linspace(sqrt(b0), sqrt(b1), N) ** 2  (fp32 like the reference Tensor)
        self.betas = (np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5,
                                  num_training_steps, dtype=np.float32) ** 2).astype(np.float32)
        self.alphas = (1.0 - self.betas).astype(np.float32)                      # :31
        self.alphas_cumprod = np.cumprod(self.alphas, dtype=np.float32)          # :32
        self.timesteps = np.arange(num_training_steps)[::-1].copy()             # :33
        self.num_inference_steps = 1
        self.start_step = 0

    def set_inference_timesteps(self, n):
        """:35-44  timesteps = round(arange(n)[::-1] * (N // n))."""
        self.num_inference_steps = n
        ratio = self.num_training_steps // n
        self.timesteps = np.round(np.arange(n)[::-1] * ratio).astype(np.int64)

    def previous_timestep(self, t):
        """:46-51."""
        return t - self.num_training_steps // self.num_inference_steps

    def variance(self, t):
        """:53-65."""
        prev = self.previous_timestep(t)
        a_t = np.float32(self.alphas_cumprod[t])
        a_prev = np.float32(self.alphas_cumprod[prev]) if prev >= 0 else np.float32(1.0)
        cur_beta = np.float32(1.0) - a_t / a_prev
        var = (np.float32(1.0) - a_prev) / (np.float32(1.0) - a_t) * cur_beta
        return np.float32(max(var, np.float32(1e-20)))

    def set_strength(self, strength):
        """:67-73 with the intended slice (App.A D21): timesteps = timesteps[start_step:]."""
        start = self.num_inference_steps - int(self.num_inference_steps * strength)
        self.timesteps = self.timesteps[start:]
        self.start_step = start

    def coefficients(self, t):
        """Scalars of `step` (:81-98): (1/sqrt(a_t), sqrt(1-a_t), c_x0, c_xt, sigma)."""
        prev = self.previous_timestep(t)
        a_t = np.float32(self.alphas_cumprod[t])
        a_prev = np.float32(self.alphas_cumprod[prev]) if prev >= 0 else np.float32(1.0)
        b_t = np.float32(1.0) - a_t
        b_prev = np.float32(1.0) - a_prev
        cur_a = a_t / a_prev
        cur_b = np.float32(1.0) - cur_a
        c_x0 = np.sqrt(a_prev) * cur_b / b_t
        c_xt = np.sqrt(cur_a) * b_prev / b_t
        sigma = np.sqrt(self.variance(t)) if t > 0 else np.float32(0.0)
        return np.float32(np.sqrt(a_t)), np.float32(np.sqrt(b_t)), np.float32(c_x0), np.float32(c_xt), np.float32(sigma)

    def step(self, t, latents, model_output, noise=None):
        """`step` :75-109: x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t); x_prev = c_x0 x0 + c_xt x (+ sigma z if t>0)."""
        sa, sb, c_x0, c_xt, sigma = self.coefficients(t)
        x0 = (latents - model_output * sb) / sa
        out = x0 * c_x0 + latents * c_xt
        if t > 0:
            out = out + noise * sigma
        return out.astype(latents.dtype)

    def add_noise(self, x, t, noise):
        """`add_noise` :111-124."""
        a = np.float32(self.alphas_cumprod[int(t)])
        return (x * np.sqrt(a) + noise * np.sqrt(np.float32(1.0) - a)).astype(x.dtype)


def cfg_combine(cond, uncond, scale):
    """pipeline.mojo:117-119: (cond - uncond) * scale + uncond."""
    return (cond - uncond) * cond.dtype.type(scale) + uncond


def denoise(P, latents, context, steps, noises, uncond_context=None, cfg_scale=7.5,
            num_training_steps=1000, sem=DEFAULT, timesteps=None):
    """The hot loop of `generate` pipeline.mojo:87-122 for one sample.

    latents (4,L,L); context (77,768); noises (steps,4,L,L) ~ N(0,1) (input, App.A D19).
    CFG (App.A D10): eps = s (eps_c - eps_u) + eps_u with eps_u from `uncond_context`."""
    s = DDPMSampler(num_training_steps)
    s.set_inference_timesteps(steps)
    ts = s.timesteps if timesteps is None else timesteps
    x = latents
    for i, t in enumerate(ts):
        temb = ops.time_embedding(float(t), sem=sem)
        eps = models.diffusion(P, x, context, temb, sem=sem)
        if uncond_context is not None:
            eps_u = models.diffusion(P, x, uncond_context, temb, sem=sem)
            eps = cfg_combine(eps, eps_u, cfg_scale)
        x = s.step(int(t), x, eps, noises[i])
    return x


def generate_image(P_unet, P_dec, latents, context, steps, noises, **kw):
    """denoise -> Decoder -> rescale (pipeline.mojo:124-127) for one sample -> (3,8L,8L) in [0,255]."""
    x = denoise(P_unet, latents, context, steps, noises, **kw)
    return ops.rescale_to_u8_range(models.decoder(P_dec, x))
