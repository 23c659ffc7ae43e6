"""Per-launch timing table of one batch-8 UNet step (and optionally the decoder): which shapes cost what."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
tsd.set_strict(True)
B, L, T = int(os.environ.get("B", 8)), int(os.environ.get("L", 64)), 77
what = os.environ.get("WHAT", "unet")
ctx = tsd.default_context()
d = tsd.Diffusion(seed=1234, variant=os.environ.get("VARIANT", "diffusion"))
dec = tsd.Decoder(seed=1234) if what == "dec" else None
enc = tsd.Encoder(seed=1234) if what == "enc" else None
lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); cx = rng.normal(1, 2, B*T*768).reshape(B,T,768)
s = tsd.Session(d.model, dec.model if dec else None, B, L, T); s.set_schedule(1000, 50, 0); s.upload(lat, cx, None, None)
if what == "enc":
    img = rng.uniform(1, 7, B*3*64*L*L, 1.0).reshape(B, 3, 8*L, 8*L); nz = rng.normal(1, 8, B*4*L*L).reshape(B, 4, L, L)
run = (lambda: s.decode()) if what == "dec" else ((lambda: enc.forward(img, nz)) if what == "enc" else (lambda: s.step(1)))
for _ in range(3): run()
ctx.synchronize()
ctx.profile_begin(); run(); recs = ctx.profile_records(); ctx.profile_end()
agg = collections.OrderedDict()
for cls, M, N, K, b, ms in recs:
    key = (cls, M, N, K, b)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ms
tot = sum(v[1] for v in agg.values())
print(f"total {tot:.3f} ms over {len(recs)} launches")
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for (cls, M, N, K, b), (n, ms) in rows[:int(os.environ.get("TOP", 45))]:
    if cls in ("gemm", "conv3x3"):
        fl = 2.0 * M * N * K * max(b, 1) * n
    elif cls == "flash_attention":
        fl = 4.0 * M * N * K * b * n
    else:
        fl = 0
    tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0
    print(f"{cls:16s} M={M:6d} N={N:5d} K={K:6d} b={b:3d} x{n:2d}  {ms:8.3f} ms  {100*ms/tot:5.1f}%  {tf:7.1f} TF")
