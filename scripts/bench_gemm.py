"""Sweep tile configurations over the UNet/VAE GEMM and conv3x3 shapes (batch 8) - tuning aid."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd._lib import lib, check
ctx = tsd.default_context()
B = 8
# (conv, H, Cin, N, stride, ups, count per step, label)
shapes = [
 (1, 64, 320, 320, 1, 0, 4, "conv L0 320->320"), (1, 64, 640, 320, 1, 0, 2, "conv L0 640->320"),
 (1, 32, 640, 640, 1, 0, 3, "conv L1 640->640"), (1, 32, 320, 640, 1, 0, 1, "conv L1 320->640"),
 (1, 32, 1280, 640, 1, 0, 1, "conv L1 1280->640 (ups src16)"), (1, 32, 960, 640, 1, 0, 1, "conv L1 960->640"),
 (1, 16, 1280, 1280, 1, 0, 3, "conv L2 1280->1280"), (1, 16, 2560, 1280, 1, 0, 1, "conv L2 2560->1280"),
 (1, 16, 1920, 1280, 1, 0, 1, "conv L2 1920->1280"), (1, 16, 640, 1280, 1, 0, 1, "conv L2 640->1280"),
 (1, 64, 320, 320, 2, 0, 1, "conv s2 320 L0->L1"), (1, 32, 640, 640, 2, 0, 1, "conv s2 640 L1->L2"),
 (0, 64, 320, 320, 1, 0, 15, "gemm L0 320x320"), (0, 64, 320, 640, 1, 0, 3, "gemm L0 qk 640"), (0, 64, 320, 2560, 1, 0, 3, "gemm L0 geglu1"),
 (0, 64, 1280, 320, 1, 0, 3, "gemm L0 geglu2"),
 (0, 32, 640, 640, 1, 0, 15, "gemm L1 640x640"), (0, 32, 640, 1280, 1, 0, 3, "gemm L1 qk"), (0, 32, 640, 5120, 1, 0, 3, "gemm L1 geglu1"),
 (0, 32, 2560, 640, 1, 0, 3, "gemm L1 geglu2"),
 (0, 16, 1280, 1280, 1, 0, 15, "gemm L2 1280x1280"), (0, 16, 1280, 2560, 1, 0, 3, "gemm L2 qk"), (0, 16, 1280, 10240, 1, 0, 3, "gemm L2 geglu1"),
 (0, 16, 5120, 1280, 1, 0, 3, "gemm L2 geglu2"),
]
if os.environ.get("VAE"):
    shapes = [(1, 64, 512, 512, 1, 0, 10, "dec conv 512 @64"), (1, 128, 512, 512, 1, 0, 7, "dec conv 512 @128"),
              (1, 256, 512, 512, 1, 0, 1, "dec conv 512 @256"), (1, 256, 256, 256, 1, 0, 5, "dec conv 256 @256"),
              (1, 512, 256, 256, 1, 0, 1, "dec conv 256 @512"), (1, 512, 128, 128, 1, 0, 5, "dec conv 128 @512")]
cfgs = [int(c) for c in os.environ.get("CFGS", "0,1,5,6,7").split(",")]
ms = C.c_float()
tot_best = 0.0
print(f"{'shape':34s} " + " ".join(f"cfg{c:>2d}(TF)" for c in cfgs) + "   best  us(best)")
for conv, H, Cin, N, stride, ups, cnt, label in shapes:
    Ho = H // stride
    fl = 2.0 * B * Ho * Ho * N * Cin * (9 if conv else 1)
    if os.environ.get("REAL_EPI"): os.environ["TSD_BENCH_EPI"] = "2" if "geglu1" in label else "1"
    res = []
    for c in cfgs:
        if N % 160 and c in (0, 1, 5, 6, 7):
            res.append(None); continue
        r = lib().tsd_debug_gemm_bench(ctx.h, conv, B, H, H, Cin, N, stride, ups, c, 20, C.byref(ms))
        res.append(ms.value if r == 0 else None)
    best = min((t for t in res if t), default=None)
    tot_best += (best or 0) * cnt
    print(f"{label:34s} " + " ".join(f"{(fl/(t*1e-3)/1e12 if t else 0):9.0f}" for t in res) + f"   cfg{cfgs[res.index(best)]}  {best*1e3:7.1f} x{cnt}")
print(f"sum over step (best cfg per shape): {tot_best:.3f} ms")
