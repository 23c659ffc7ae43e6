"""K sweep of one conv problem (C -> 320 at 64 x 64, batch 8: M = 32768) over tile configurations: launch time at eight K lengths
(profiles/README.md round 5: T = 14.5 + 1.0 x K-tiles us for cfg 51 - every tile runs ~1.3 PF inside its K loop).  python scripts/kloop_sweep.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "stable-diffusion.mojo_amd"))
os.environ.setdefault("TSD_BENCH_EPI", "1")
import tsd
from tsd._lib import lib
ctx = tsd.Context(0)
for cfg in (51, 11, 0, 5, 45):
    row=[]
    for cin in (64, 128, 192, 320, 448, 640, 960, 1280):
        ms = C.c_float()
        r = lib().tsd_debug_gemm_bench(ctx.h, 1, 8, 64, 64, cin, 320, 1, 0, cfg, 30, C.byref(ms))
        row.append((9*cin//64, round(ms.value*1e3,1)) if r==0 else None)
    print("cfg", cfg, row)
