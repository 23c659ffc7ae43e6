"""Cross-check every tile configuration against a reference configuration on a few shapes (GPU)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context(); d = C.c_float(); m = C.c_float()
cfgs = [int(c) for c in os.environ.get("CFGS", "16,17").split(",")]
shapes = [(1, 2, 16, 320, 320, 1, 0), (1, 8, 64, 320, 320, 1, 0), (1, 8, 32, 640, 640, 1, 0), (1, 2, 16, 320, 320, 2, 0), (1, 2, 8, 640, 320, 1, 1),
          (0, 8, 64, 320, 320, 1, 0), (0, 8, 16, 1280, 1280, 1, 0), (0, 1, 8, 320, 160, 1, 0), (0, 3, 8, 64, 320, 1, 0), (0, 8, 32, 640, 5120, 1, 0)]
bad = 0
for conv, B, H, Cin, N, stride, ups in shapes:
    for c in cfgs:
        if N % 160 and c in (0, 1, 5, 6, 7, 11, 12, 14, 15, 16, 17): continue
        r = lib().tsd_debug_gemm_check(ctx.h, conv, B, H, H, Cin, N, stride, ups, c, 1 if N % 160 == 0 else 3, C.byref(d), C.byref(m))
        ok = r == 0 and d.value == 0.0
        bad += not ok
        print(f"conv={conv} B={B} H={H} Cin={Cin} N={N} s={stride} ups={ups} cfg={c}: rc={r} max|diff|={d.value:.4g} max|ref|={m.value:.3g} {'OK' if ok else 'MISMATCH'}")
print("MISMATCHES:", bad)
