"""halo-x conv configurations (30 = 128x160, 32 = 128x128) against the plain ones (0 / 2) on the shapes they are eligible for."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context()
d, r = C.c_float(), C.c_float()
ms = C.c_float()
for (B, H, Cin, N, cfg, ref) in [(2, 64, 64, 160, 30, 0), (8, 64, 320, 320, 30, 0), (2, 64, 640, 320, 30, 0), (1, 128, 128, 128, 32, 2), (2, 128, 64, 160, 30, 0),
                                 (1, 256, 128, 128, 32, 2), (2, 64, 128, 256, 32, 2)]:
    rc = lib().tsd_debug_gemm_check(ctx.h, 1, B, H, H, Cin, N, 1, 0, cfg, ref, C.byref(d), C.byref(r))
    print(f"B={B} H=W={H} Cin={Cin} N={N} cfg {cfg} vs {ref}: rc={rc} max|diff|={d.value:.5f} max|ref|={r.value:.3f}")
if os.environ.get("TIME"):
    for (B, H, Cin, N) in [(8, 64, 320, 320), (8, 64, 640, 320), (8, 128, 512, 512), (8, 256, 256, 256), (8, 512, 128, 128)]:
        row = []
        for cfg in ((0, 30, 0, 30) if N % 160 == 0 else (2, 32, 2, 32)):
            lib().tsd_debug_gemm_bench(ctx.h, 1, B, H, H, Cin, N, 1, 0, cfg, 10, C.byref(ms))
            row.append(f"{cfg}:{ms.value*1e3:8.1f}us")
        print(f"B={B} H={H} Cin={Cin} N={N}  " + "  ".join(row))
