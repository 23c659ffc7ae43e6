"""Decoder-sized conv3x3 shapes: the 128x128 two-blocks-per-CU tile (cfg 2) against the 8-wave 256x128 tile (cfg 13), alternating."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
def t(conv, B, H, K, N, cfg, iters=5):
    r = lib().tsd_debug_gemm_bench(ctx.h, conv, B, H, H, K, N, 1, 0, cfg, iters, C.byref(ms))
    return ms.value * 1e3 if r == 0 else float("nan")
for (conv, B, H, K, N) in [(1, 8, 512, 128, 128), (1, 8, 512, 256, 128), (1, 8, 256, 256, 256), (1, 8, 256, 512, 256), (1, 8, 128, 512, 512), (1, 8, 256, 512, 512), (1, 8, 512, 256, 256), (0, 8, 512, 256, 128)]:
    fl = 2.0 * B * H * H * N * K * (9 if conv else 1)
    row = []
    for cfg in (2, 13, 2, 13, -1):
        us = t(conv, B, H, K, N, cfg)
        row.append(f"{cfg}:{us:7.1f}us {fl/us/1e6:6.0f}TF")
    print(f"conv={conv} M={B*H*H} N={N} K={K*(9 if conv else 1)}  " + "  ".join(row))
