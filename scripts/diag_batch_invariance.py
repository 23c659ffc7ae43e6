"""Bitwise batch invariance of Diffusion.forward at the headline size for several batch sizes (incl. odd ones)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
u = tsd.Diffusion(seed=1234)
L, T, Bmax = 64, 77, 8
lat = tsd.rng.normal(5, 1, Bmax * 4 * L * L).reshape(Bmax, 4, L, L)
ctx = tsd.rng.normal(5, 2, Bmax * T * 768).reshape(Bmax, T, 768)
temb = np.stack([tsd.get_time_embedding(float(t)).reshape(320) for t in (980, 700, 500, 300, 100, 60, 20, 0)])
ref = u.forward(lat, ctx, temb)
for B in (1, 2, 3, 5, 7, 8):
    out = u.forward(lat[:B], ctx[:B], temb[:B])
    same = np.array_equal(out, ref[:B])
    print(f"B={B}: bitwise equal to the batch-8 rows: {same}; max|diff| = {np.abs(out - ref[:B]).max():.3e}")
print("split-K hand-off errors:", lib().tsd_debug_splitk_errors(tsd.default_context().h))
