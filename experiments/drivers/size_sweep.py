"""Edge-size sweep on the GPU: the UNet and the decoder at odd batch / latent sizes - finite outputs, bitwise batch
invariance against single-sample calls, loud errors (never silent corruption) where a size is unsupported."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd import rng
tsd.set_strict(True)
d = tsd.Diffusion(seed=1234)
dec = tsd.Decoder(seed=1234)
T = 77
for (B, L) in [(1, 8), (3, 24), (5, 40), (2, 96), (16, 64), (32, 64), (1, 128), (7, 16), (2, 12)]:
    try:
        lat = rng.normal(1, 10 + L, B * 4 * L * L).reshape(B, 4, L, L)
        cx = rng.normal(1, 20 + L, B * T * 768).reshape(B, T, 768)
        te = np.stack([tsd.get_time_embedding(float(37 * (b + 1) % 1000)).reshape(320) for b in range(B)])
        t0 = time.time(); y = d.forward(lat, cx, te); dt = time.time() - t0
        ok = bool(np.isfinite(y).all())
        j = B - 1
        y1 = d.forward(lat[j], cx[j], te[j])
        same = bool(np.array_equal(y1, y[j]))
        msg = f"unet B={B:2d} L={L:3d}: finite={ok} std={y.std():.4f} bitwise_batch_invariant={same} ({dt*1e3:.0f} ms incl. PCIe)"
        if B * L * L <= 16 * 64 * 64 and L <= 64:
            img = dec.forward(lat[:min(B, 4)] if lat.ndim == 4 else lat)
            msg += f" | decoder finite={bool(np.isfinite(img).all())} shape={img.shape}"
        print(msg, flush=True)
    except tsd.TsdError as e:
        print(f"unet B={B} L={L}: TsdError {e}", flush=True)
print("splitk errors", tsd._lib.lib().tsd_debug_splitk_errors(tsd.default_context().h))
