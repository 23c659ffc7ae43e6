import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
tsd.set_strict(True)
d = tsd.Diffusion(seed=1234)
B, L = 1, 8
lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); ctx = rng.normal(1, 2, B*77*768).reshape(B,77,768); uctx = rng.normal(1, 3, B*77*768).reshape(B,77,768)
te = tsd.get_time_embedding(500.0).reshape(1,320)
e_c = d.forward(lat, ctx, te); e_u = d.forward(lat, uctx, te)
for scale in (1.0, 7.5):
    s = tsd.Session(d.model, None, B, L, 77, cfg=True); s.set_schedule(1000, 2, 0)
    s.upload(lat, ctx, uctx, None, cfg_scale=scale); s.step(0); got = s.latents(); s.close()
    sm = tsd.DDPMSampler(0, 1000); sm.set_inference_timesteps(2)
    a_t = sm.alphas_cumprod[500]; a_p = sm.alphas_cumprod[0]
    eps = (e_c - e_u) * np.float32(scale) + e_u
    cur_a = a_t / a_p
    x0 = (lat - eps * np.sqrt(1 - a_t)) / np.sqrt(a_t)
    ref = x0 * (np.sqrt(a_p) * (1 - cur_a) / (1 - a_t)) + lat * (np.sqrt(cur_a) * (1 - a_p) / (1 - a_t))
    print("scale", scale, "rel", np.linalg.norm(got-ref)/np.linalg.norm(ref))
# no-cfg session with cond ctx only
s = tsd.Session(d.model, None, B, L, 77, cfg=False); s.set_schedule(1000, 2, 0)
s.upload(lat, ctx, None, None); s.step(0); got = s.latents(); s.close()
x0 = (lat - e_c * np.sqrt(1 - a_t)) / np.sqrt(a_t)
ref = x0 * (np.sqrt(a_p) * (1 - cur_a) / (1 - a_t)) + lat * (np.sqrt(cur_a) * (1 - a_p) / (1 - a_t))
print("nocfg rel", np.linalg.norm(got-ref)/np.linalg.norm(ref))
