"""Persistent tile loop experiment (-DTSD_GEMM_PT build, TSD_LIB): bitwise check of the 256x128 staggered conv tile walking many tiles per workgroup against the
128x128 tile, then decode / encode timing with the loop on and off (TSD_GEMM_PT_OFF=1)."""
import os, sys, ctypes as C, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import numpy as np
import tsd
from tsd import rng
from tsd._lib import lib
ctx = tsd.default_context()
d, m, ms = C.c_float(), C.c_float(), C.c_float()
if os.environ.get("CHECK"):
    for (B, H, Cin, N) in [(2, 256, 128, 256), (1, 256, 256, 512), (2, 128, 512, 512), (8, 64, 512, 512)]:
        r = lib().tsd_debug_gemm_check(ctx.h, 1, B, H, H, Cin, N, 1, 0, 53, 2, C.byref(d), C.byref(m))
        print(f"check conv B={B} H={H} Cin={Cin} N={N}: rc={r} max|diff|={d.value} max|ref|={m.value}")
    for (B, H, Cin, N) in [(2, 256, 256, 256), (8, 128, 512, 512), (8, 256, 256, 256)]:
        r = lib().tsd_debug_gemm_bench(ctx.h, 1, B, H, H, Cin, N, 1, 0, 53, 10, C.byref(ms))
        fl = 2.0 * B * H * H * N * 9 * Cin
        print(f"bench conv B={B} H={H} Cin={Cin} N={N} cfg 53: {ms.value*1e3:8.1f} us {fl/(ms.value*1e-3)/1e12:7.1f} TF rc={r}")
dec = tsd.Decoder(seed=1234); enc = tsd.Encoder(seed=1234)
lat = rng.normal(1, 1, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)
un = tsd.Diffusion(seed=1234)
s = tsd.Session(un.model, dec.model, 8, 64, 77); s.set_schedule(1000, 50, 0); s.upload(lat, rng.normal(1, 2, 8 * 77 * 768).reshape(8, 77, 768), None, None)
for _ in range(3): s.decode()
ctx.synchronize(); t0 = time.time()
for _ in range(10): s.decode()
ctx.synchronize(); dt = (time.time() - t0) / 10
img = s.images(rescale=False)
print(f"decode 8 x 512^2: {dt*1e3:.2f} ms  finite={np.isfinite(img).all()} checksum={float(np.abs(img).sum()):.6e}")
