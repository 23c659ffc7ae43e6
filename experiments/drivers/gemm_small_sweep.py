"""Time the small dense GEMMs of the 32x32 / 16x16 attention blocks across tile configurations and K (fixed-cost floor)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
def t(H, K, N, cfg, iters=50):
    r = lib().tsd_debug_gemm_bench(ctx.h, 0, 8, H, H, K, N, 1, 0, cfg, iters, C.byref(ms))
    return ms.value * 1e3 if r == 0 else float("nan")
for (H, K, N) in [(32, 640, 640), (16, 1280, 1280), (32, 32, 640), (32, 64, 640), (32, 160, 640), (32, 320, 640), (32, 1280, 640), (32, 2560, 640), (16, 32, 1280), (16, 640, 1280), (16, 2560, 1280)]:
    row = []
    for cfg in (-1, 0, 1, 5, 6, 7, 11):
        row.append(f"{cfg}:{t(H, K, N, cfg):6.1f}")
    fl = 2.0 * 8 * H * H * N * K
    print(f"M={8*H*H:5d} N={N:4d} K={K:4d}  " + "  ".join(row) + f"   ({fl/1e9:.2f} GF)")
