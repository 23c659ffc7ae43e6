"""256x320 tile with every CU busy: what a split-K slice of it would cost (a slice = a launch with 1/S of the K at S times the batch)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("TSD_BENCH_WROT", "4")
import tsd
from tsd._lib import lib, Context
ctx = Context(0)
ms = C.c_float()
# (B, H, Cin, N, label): M = B*H*H rows, tiles = M/256 * N/320
rows = [(16, 64, 320, 320, "L0 320->320 x2 batch  = a 2-way slice of 640->320 (45 K tiles, 256 WG)"),
        (16, 64, 192, 320, "L0 192->320 x2 batch  ~ a 2-way slice of 320->320 (27 K tiles)"),
        (16, 64, 448, 320, "L0 448->320 x2 batch  ~ a 2-way slice of 960->320 (63 K tiles)"),
        (32, 32, 192, 640, "L1 192->640 x4 batch  ~ a 4-way slice of 640->640 (27 K tiles)"),
        (32, 32, 320, 640, "L1 320->640 x4 batch  = a 4-way slice of 1280->640 (45 K tiles)"),
        (64, 16, 192, 1280, "L2 192->1280 x8 batch ~ an 8-way slice of 1280->1280 (27 of 22.5 K tiles)"),
        (64, 16, 320, 1280, "L2 320->1280 x8 batch = an 8-way slice of 2560->1280 (45 K tiles)"),
        (32, 16, 320, 1280, "L2 320->1280 x4 batch = a 4-way slice of 1280->1280 (45 K tiles, 128 WG)")]
for B, H, Cin, N, label in rows:
    out = []
    for c in (56, 51, -1):
        r = lib().tsd_debug_gemm_bench(ctx.h, 1, B, H, H, Cin, N, 1, 0, c, 30, C.byref(ms))
        fl = 2.0 * B * H * H * N * Cin * 9
        out.append(f"cfg{c:>3d} {ms.value * 1e3:7.1f} us {fl / (ms.value * 1e-3) / 1e12:5.0f} TF" if r == 0 else f"cfg{c} fail")
    print(f"{label:80s} " + " | ".join(out), flush=True)
