"""Host-side mirror of `sampler.mojo` (DDPMSampler).  The schedule scalars are host code exactly as in the
reference (a few fp32 scalars per step); the per-step tensor update runs on the GPU inside the
session (`tsd_session_step`: fused CFG combine + posterior mean + noise, SURVEY.md App.D K9)."""
import numpy as np


class DDPMSampler:
    """`DDPMSampler` sampler.mojo:5-124 (num_training_steps is a parameter: App.A D22)."""

    def __init__(self, seed_val=0, num_training_steps=1000, beta_start=0.00085, beta_end=0.0120):
        self.seed_val = seed_val
        self.num_training_steps = num_training_steps
        self.betas = (np.linspace(np.float32(beta_start) ** 0.5, np.float32(beta_end) ** 0.5, num_training_steps,
                                  dtype=np.float32) ** 2).astype(np.float32)          # :28-30
        self.alphas = (1.0 - self.betas).astype(np.float32)                           # :31
        self.alphas_cumprod = np.cumprod(self.alphas, dtype=np.float32)               # :32
        self.timesteps = np.arange(num_training_steps)[::-1].copy()                  # :33
        self.num_inference_steps = 1
        self.start_step = 0

    def set_inference_timesteps(self, num_inference_steps=1):                         # :35-44
        self.num_inference_steps = num_inference_steps
        ratio = self.num_training_steps // num_inference_steps
        self.timesteps = np.round(np.arange(num_inference_steps)[::-1] * ratio).astype(np.int64)

    def get_previous_timestep(self, timestep):                                        # :46-51
        return timestep - self.num_training_steps // self.num_inference_steps

    def get_variance(self, timestep):                                                 # :53-65
        prev = self.get_previous_timestep(timestep)
        a_t = np.float32(self.alphas_cumprod[timestep])
        a_prev = np.float32(self.alphas_cumprod[prev]) if prev >= 0 else np.float32(1.0)
        cur_beta = np.float32(1.0) - a_t / a_prev
        var = (np.float32(1.0) - a_prev) / (np.float32(1.0) - a_t) * cur_beta
        return np.float32(max(var, np.float32(1e-20)))

    def set_strength(self, strength):                                                 # :67-73 (intended slice, App.A D21)
        start = self.num_inference_steps - int(self.num_inference_steps * strength)
        self.timesteps = self.timesteps[start:]
        self.start_step = start
