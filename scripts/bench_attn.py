"""Time the fused attention core at the UNet's shapes (batch 8)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); ms = C.c_float()
for (d, S, Sk) in [(40, 4096, 4096), (80, 1024, 1024), (160, 256, 256), (40, 4096, 77), (80, 1024, 77), (160, 256, 77)]:
    r = lib().tsd_debug_attn_bench(ctx.h, 8, 8, d, S, Sk, int(os.environ.get("ITERS", 10)), C.byref(ms))
    fl = 4.0 * 64 * S * Sk * d
    print(f"d={d:3d} Sq={S:5d} Sk={Sk:5d}: {ms.value*1e3:8.1f} us  {fl/(ms.value*1e-3)/1e12:6.1f} TF (algorithmic)  rc={r}")
