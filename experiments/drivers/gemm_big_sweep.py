"""Big conv3x3 / GEMM shapes of the UNet and the decoder across tile configurations (needs -DTSD_GEMM_EXPERIMENTAL for 14-21)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
def t(conv, B, H, K, N, cfg, iters=20):
    r = lib().tsd_debug_gemm_bench(ctx.h, conv, B, H, H, K, N, 1, 0, cfg, iters, C.byref(ms))
    return ms.value * 1e3 if r == 0 else float("nan")
CF = [int(x) for x in os.environ.get("CFGS", "-1,5,11,14,16").split(",")]
for (conv, B, H, K, N) in [(1, 8, 64, 320, 320), (1, 8, 64, 640, 320), (1, 8, 32, 640, 640), (1, 8, 16, 1280, 1280), (1, 8, 128, 512, 512), (1, 8, 256, 256, 256),
                           (0, 8, 32, 640, 5120), (0, 8, 16, 1280, 10240)]:
    fl = 2.0 * B * H * H * N * K * (9 if conv else 1)
    row = []
    for cfg in CF:
        us = t(conv, B, H, K, N, cfg)
        row.append(f"{cfg}:{us:7.1f}us {fl/us/1e6:6.0f}TF")
    print(f"conv={conv} M={B*H*H:7d} N={N:5d} K={K*(9 if conv else 1):5d}  " + "  ".join(row))
