"""Which encoder call reads memory nobody wrote?  (TSD_DEBUG_POISON set by the caller.)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tsd
from tsd import rng, checkpoint as ck
from oracle import spec
SEED = 1234
order = sys.argv[1] if len(sys.argv) > 1 else "direct,loaded"
variant = sys.argv[2] if len(sys.argv) > 2 else "encoder_torch"
img = rng.uniform(SEED, 591, 3 * 64 * 64, 1.0).reshape(3, 64, 64)
nz = rng.normal(SEED, 592, 4 * 8 * 8).reshape(4, 8, 8)
res = {}
for what in order.split(","):
    try:
        if what == "direct":
            m = tsd.Encoder(seed=SEED, variant=variant)
        else:
            Pe = spec.init_params(variant, SEED)
            m = ck.load_vae(ck.params_to_diffusers_vae(Pe, "encoder"), "encoder")
        y = np.asarray(m.forward(img, nz)); res[what] = y
        print(what, "ok finite", bool(np.isfinite(y).all()), "sum", float(y.astype(np.float64).sum()))
    except Exception as e:
        print(what, "ERROR", str(e)[:120])
if len(res) == 2:
    a, b = res.values(); print("equal", bool(np.array_equal(a, b)), "max diff", float(np.abs(a - b).max()))
