import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
d, S = int(os.environ.get("D", 40)), int(os.environ.get("S", 4096))
r = lib().tsd_debug_attn_bench(ctx.h, 8, 8, d, S, S, 3, C.byref(ms))
print(d, S, ms.value * 1e3, "us", r)
