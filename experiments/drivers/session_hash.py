"""sha256 of the latents after N session steps at the headline size (A/B of session-level changes, e.g. TSD_TIME_OVERLAP)."""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
tsd.set_strict(True)
B, L, T, n = 8, 64, 77, 50
cfg = os.environ.get("CFG", "0") == "1"
d = tsd.Diffusion(seed=1234)
lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); cx = rng.normal(1, 2, B*T*768).reshape(B,T,768); ux = rng.normal(1, 4, B*T*768).reshape(B,T,768)
nz = rng.normal(1, 3, n*B*4*L*L).reshape(n,B,4,L,L)
s = tsd.Session(d.model, None, B, L, T, cfg=cfg); s.set_schedule(1000, n, 0); s.upload(lat, cx, ux if cfg else None, nz)
for i in range(n): s.step(i)
out = s.latents()
print("overlap", os.environ.get("TSD_TIME_OVERLAP", "1"), "cfg", cfg, "sha256", hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16], "finite", bool(np.isfinite(out).all()))
# out-of-order steps must still be right: repeat step 7 twice, then 3
s.upload(lat, cx, ux if cfg else None, nz); s.step(7); a = s.latents(); s.upload(lat, cx, ux if cfg else None, nz); s.step(3); s.upload(lat, cx, ux if cfg else None, nz); s.step(7); b = s.latents()
print("out-of-order equal:", bool(np.array_equal(a, b)))
s.close()
