"""Scalar-path L2 prefetch experiment (-DTSD_GEMM_SPF builds): the 16x16-level convs (weights beyond the L2) on the 128x160 loader-wave tile (cfg 45),
with the weights of every launch cold (TSD_BENCH_WROT=40) or the same every launch (1)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
for (conv, H, cin, N, cfg) in [(1, 16, 1280, 1280, 45), (1, 16, 2560, 1280, 45), (0, 16, 5120, 1280, 45), (0, 16, 1280, 1280, 47), (0, 32, 640, 640, 45)]:
    r = lib().tsd_debug_gemm_bench(ctx.h, conv, 8, H, H, cin, N, 1, 0, cfg, 20, C.byref(ms))
    K = 9 * cin if conv else cin
    fl = 2.0 * 8 * H * H * N * K
    print(f"lib={os.path.basename(os.environ.get('TSD_LIB') or 'shipped'):18s} wrot={os.environ.get('TSD_BENCH_WROT','1'):3s} conv={conv} H={H} cin={cin} N={N} cfg={cfg}: {ms.value*1e3:8.1f} us {fl/(ms.value*1e-3)/1e12:7.1f} TF rc={r}")
