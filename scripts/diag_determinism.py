"""GPU diagnostic: run-to-run determinism and batch invariance of Diffusion.forward (bitwise)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
tsd.set_strict(True)
d = tsd.Diffusion(seed=1234)
for L in (8, 16, 32):
    B = 2
    lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); ctx = rng.normal(1, 2, B*77*768).reshape(B,77,768)
    lat[1] = lat[0]
    te = np.stack([tsd.get_time_embedding(500.0).reshape(320)]*B)
    a = d.forward(lat, ctx, te); b = d.forward(lat, ctx, te)
    s0 = d.forward(lat[0], ctx[0], te[0]); s1 = d.forward(lat[1], ctx[1], te[1])
    print(f"L={L}: same-call-twice max|d|={np.abs(a-b).max():.3e}  batched-vs-single: {np.abs(a[0]-s0).max():.3e} {np.abs(a[1]-s1).max():.3e}  |out|max={np.abs(a).max():.3f}  cond-vs-uncond diff rel={np.linalg.norm(a[0]-a[1])/np.linalg.norm(a[0]):.3e}")
