"""256x320 tile (configuration 56) against the tiles the dispatcher picks today: bitwise check, then repeated-launch timing with
cold weights (TSD_BENCH_WROT) and the real epilogue, on the shapes whose N is a multiple of 320."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("TSD_BENCH_WROT", "4")
import tsd
from tsd._lib import lib, Context
B = 8
# (conv, H, Cin, N, stride, epi, label)
shapes = [
 (1, 64, 320, 320, 1, 1, "conv L0 320->320"), (1, 64, 640, 320, 1, 1, "conv L0 640->320"), (1, 64, 960, 320, 1, 1, "conv L0 960->320"),
 (1, 32, 640, 640, 1, 1, "conv L1 640->640"), (1, 32, 1280, 640, 1, 1, "conv L1 1280->640"),
 (1, 16, 1280, 1280, 1, 1, "conv L2 1280->1280"), (1, 16, 2560, 1280, 1, 1, "conv L2 2560->1280"),
 (0, 64, 320, 640, 1, 0, "gemm L0 qk 640"), (0, 64, 320, 2560, 1, 2, "gemm L0 geglu1"),
 (0, 32, 640, 640, 1, 1, "gemm L1 640x640"), (0, 32, 640, 1920, 1, 0, "gemm L1 qkv"), (0, 32, 640, 5120, 1, 2, "gemm L1 geglu1"),
 (0, 32, 2560, 640, 1, 1, "gemm L1 geglu2"),
 (0, 16, 1280, 1280, 1, 1, "gemm L2 1280x1280"), (0, 16, 1280, 3840, 1, 0, "gemm L2 qkv"), (0, 16, 1280, 10240, 1, 2, "gemm L2 geglu1"),
 (0, 16, 5120, 1280, 1, 1, "gemm L2 geglu2"),
]
cfgs = [int(c) for c in os.environ.get("CFGS", "-1,0,51,56").split(",")]
ctxs = {}
for e in (0, 1, 2):
    os.environ["TSD_BENCH_EPI"] = str(e)
    ctxs[e] = Context(0)
ms = C.c_float(); md = C.c_float(); mr = C.c_float()
print(f"{'shape':28s} check56 " + " ".join(f"cfg{c:>3d} us / TF" for c in cfgs))
for conv, H, Cin, N, stride, epi, label in shapes:
    Ho = H // stride
    fl = 2.0 * B * Ho * Ho * N * Cin * (9 if conv else 1)
    r = lib().tsd_debug_gemm_check(ctxs[0].h, conv, B, H, H, Cin, N, stride, 0, 56, 0, C.byref(md), C.byref(mr))
    chk = f"{md.value:.1e}" if r == 0 else f"rc{r}"
    out = []
    for c in cfgs:
        r = lib().tsd_debug_gemm_bench(ctxs[epi].h, conv, B, H, H, Cin, N, stride, 0, c, 30, C.byref(ms))
        out.append(f"{ms.value * 1e3:7.1f} {fl / (ms.value * 1e-3) / 1e12:5.0f}" if r == 0 else "   fail     ")
    print(f"{label:28s} {chk:>7s} " + "  ".join(out), flush=True)
