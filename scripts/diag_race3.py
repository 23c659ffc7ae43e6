"""Determinism stress of the session path (what test_img2img_config4_full_size_properties runs): N x generate() of 8 images -
img2img through the encoder (STEPS UNet steps) - on the same inputs; counts the distinct results.  After each call the latents
of every step could differ only through a race or an uninitialised read."""
import os, sys, hashlib, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd import rng
SEED = 1234
B, L = int(os.environ.get("B", 8)), 64
d = tsd.Diffusion(seed=SEED); dec = tsd.Decoder(seed=SEED); enc = tsd.Encoder(seed=SEED)
ctx = rng.normal(SEED, 761, B * 77 * 768).reshape(B, 77, 768)
image = rng.uniform(SEED, 761, B * 3 * 512 * 512, 1.0).reshape(B, 3, 512, 512) * 127.5 + 127.5
N = int(os.environ.get("N", 10))
mode = os.environ.get("MODE", "img2img")
kw = dict(cfg=False, inference_steps=50, seed_val=31, L=L)
if mode == "img2img": kw.update(input_image=image, encoder=enc, strength=0.6)
cnt = collections.Counter(); outs = {}
for i in range(N):
    o = tsd.generate(d, dec, ctx, **kw)
    h = hashlib.sha1(o.tobytes()).hexdigest()[:10]
    cnt[h] += 1; outs.setdefault(h, (i, o))
print(f"{mode}: {N} x generate: {len(cnt)} distinct results {dict(cnt)} first seen at {[v[0] for v in outs.values()]}")
if len(cnt) > 1:
    import itertools
    (ia, a), (ib, b) = list(outs.values())[:2]
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    per = d.reshape(d.shape[0], -1).max(axis=1)
    print("  max |diff| per image:", np.round(per, 4).tolist(), "pixels differing:", int((d > 0).sum()), "of", d.size)
