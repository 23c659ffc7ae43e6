"""Run-to-run determinism stress: N batch-8 Diffusion.forward calls at the headline size on the same inputs; counts the distinct outputs
(sha1 of the bytes) and, for the outliers, where they differ.  usage: N=60 python scripts/diag_race.py"""
import os, sys, hashlib, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
u = tsd.Diffusion(seed=1234)
L, T, B = int(os.environ.get("L", 64)), 77, int(os.environ.get("B", 8))
lat = tsd.rng.normal(5, 1, B * 4 * L * L).reshape(B, 4, L, L)
ctx = tsd.rng.normal(5, 2, B * T * 768).reshape(B, T, 768)
temb = np.stack([tsd.get_time_embedding(float(t)).reshape(320) for t in (980, 700, 500, 300, 100, 60, 20, 0)][:B])
N = int(os.environ.get("N", 60))
outs, cnt = {}, collections.Counter()
for i in range(N):
    o = u.forward(lat, ctx, temb)
    h = hashlib.sha1(o.tobytes()).hexdigest()[:10]
    cnt[h] += 1
    outs.setdefault(h, o)
print(f"{N} forwards, B={B}, L={L}: {len(cnt)} distinct outputs {dict(cnt)}")
if len(cnt) > 1:
    ref = outs[cnt.most_common(1)[0][0]]
    for h, o in outs.items():
        d = np.abs(o - ref)
        if d.max() > 0:
            bad = np.argwhere(d.reshape(B, -1).max(axis=1) > 0).ravel()
            print(f"  {h}: max|diff| {d.max():.3e}, samples differing {bad.tolist()}, elements {int((d > 0).sum())}")
