"""Linear + LayerNorm as one kernel (kernels_rowln.hip, C = 640) against the GEMM + LayerNorm pair it replaces."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
ctx = tsd.default_context()
ms = (C.c_float * 2)()
for M in (8192, 4096, 16384):
    for res in (0, 1):
        r = lib().tsd_debug_row_ln_bench(ctx.h, M, res, 2, 1, ms)
        d = (ms[0], ms[1])
        r0 = lib().tsd_debug_row_ln_bench(ctx.h, M, res, 0, 50, ms); t0 = ms[0] * 1e3
        r1 = lib().tsd_debug_row_ln_bench(ctx.h, M, res, 1, 50, ms); t1 = ms[0] * 1e3
        print(f"M={M:6d} residual={res}: fused {t0:6.1f} us   gemm+ln {t1:6.1f} us   max|diff| tok {d[0]:.4f} ln {d[1]:.4f}  rc={r},{r0},{r1}")
