"""A/B of the loader-wave tile configurations (45-53) against the dispatcher's choice on the UNet / VAE GEMM and conv3x3 shapes."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd._lib import lib
ctx = tsd.default_context()
B = 8
shapes = [  # (conv, H, Cin, N, stride, label)
 (1, 64, 320, 320, 1, "conv L0 320->320"), (1, 64, 640, 320, 1, "conv L0 640->320"),
 (1, 32, 640, 640, 1, "conv L1 640->640"), (1, 32, 320, 640, 1, "conv L1 320->640"), (1, 32, 1280, 640, 1, "conv L1 1280->640"), (1, 32, 960, 640, 1, "conv L1 960->640"),
 (1, 16, 1280, 1280, 1, "conv L2 1280->1280"), (1, 16, 2560, 1280, 1, "conv L2 2560->1280"), (1, 16, 640, 1280, 1, "conv L2 640->1280"),
 (0, 32, 640, 640, 1, "gemm L1 640x640"), (0, 32, 640, 1280, 1, "gemm L1 qk"), (0, 32, 640, 5120, 1, "gemm L1 geglu1"), (0, 32, 2560, 640, 1, "gemm L1 geglu2"),
 (0, 16, 1280, 1280, 1, "gemm L2 1280x1280"), (0, 16, 1280, 2560, 1, "gemm L2 qk"), (0, 16, 1280, 10240, 1, "gemm L2 geglu1"), (0, 16, 5120, 1280, 1, "gemm L2 geglu2"),
]
if os.environ.get("VAE"):
    shapes = [(1, 64, 512, 512, 1, "dec conv 512 @64"), (1, 128, 512, 512, 1, "dec conv 512 @128"), (1, 256, 256, 256, 1, "dec conv 256 @256"),
              (1, 512, 128, 128, 1, "dec conv 128 @512")]
cfgs = [int(c) for c in os.environ.get("CFGS", "-1,0,5,45,7,47,6,46,11,51").split(",")]
ms = C.c_float()
print(f"{'shape':28s} " + " ".join(f"cfg{c:>3d}" for c in cfgs) + "   (us; -1 = dispatcher)")
for conv, H, Cin, N, stride, label in shapes:
    os.environ["TSD_BENCH_EPI"] = "2" if "geglu1" in label else "1"
    res = []
    for c in cfgs:
        best = 1e9
        for rep in range(3):
            r = lib().tsd_debug_gemm_bench(ctx.h, conv, B, H, H, Cin, N, stride, 0, c, 20, C.byref(ms))
            if r != 0: best = float("nan"); break
            best = min(best, ms.value * 1e3)
        res.append(best)
    print(f"{label:28s} " + " ".join(f"{v:6.1f}" for v in res))
