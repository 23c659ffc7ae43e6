import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
conv, H, Cin, N, cfg = [int(x) for x in os.environ.get("SHAPE", "1,64,640,320,0").split(",")]
r = lib().tsd_debug_gemm_bench(ctx.h, conv, 8, H, H, Cin, N, 1, 0, cfg, 20, C.byref(ms))
fl = 2.0 * 8 * H * H * N * Cin * (9 if conv else 1)
print(conv, H, Cin, N, cfg, ms.value * 1e3, "us", fl / (ms.value * 1e-3) / 1e12, "TF", r)
