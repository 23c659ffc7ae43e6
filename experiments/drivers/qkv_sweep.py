import os, sys, ctypes as C
ROOT = "/root/repo" if os.path.isdir("/root/repo/stable-diffusion.mojo_amd") else os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd._lib import lib
ctx = tsd.default_context()
ms = C.c_float()
os.environ["TSD_BENCH_EPI"] = "0"
for label, H, Cin, N in [("L1 qk 1280", 32, 640, 1280), ("L1 qkv 1920", 32, 640, 1920), ("L2 qk 2560", 16, 1280, 2560), ("L2 qkv 3840", 16, 1280, 3840)]:
    row = []
    for c in [-1, 0, 1, 5, 7, 11, 51]:
        r = lib().tsd_debug_gemm_bench(ctx.h, 0, 8, H, H, Cin, N, 1, 0, c, 30, C.byref(ms))
        row.append(f"cfg{c}: {ms.value*1e3:6.1f}" if r == 0 else f"cfg{c}: err")
    print(f"{label:14s} " + "  ".join(row))
