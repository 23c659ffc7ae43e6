"""sha256 of one batch-8 Diffusion.forward at the headline size (bitwise A/B between two builds of libtsd.so)."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
tsd.set_strict(True)
B, L, T = int(os.environ.get("B", 8)), int(os.environ.get("L", 64)), 77
d = tsd.Diffusion(seed=1234)
lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); cx = rng.normal(1, 2, B*T*768).reshape(B,T,768)
temb = np.stack([tsd.get_time_embedding(float(t)).reshape(320) for t in np.linspace(980, 0, B)])
out = d.forward(lat, cx, temb)
print("forward sha256", hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16], "finite", bool(np.isfinite(out).all()), "std", float(out.std()))
