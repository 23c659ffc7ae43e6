"""K-loop slope / fixed cost of chosen tile configurations: time = fixed + slope x K-tiles from two K lengths of a one-tile-per-CU launch
(the measurement behind bench.py's roofline.k_loop_model, for any configuration list).
   python scripts/kloop_probe.py "51,11,5,45" [conv|dense]      env: TSD_BENCH_EPI=1 (bias + residual epilogue), ITERS"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
os.environ.setdefault("TSD_BENCH_EPI", "1")
import tsd
from tsd._lib import lib
ctx = tsd.Context(0)
cfgs = [int(c) for c in (sys.argv[1] if len(sys.argv) > 1 else "51,11,5").split(",")]
kind = sys.argv[2] if len(sys.argv) > 2 else "conv"
iters = int(os.environ.get("ITERS", "30"))
# (label, conv, B, H, W, N, (Cin short, long))
probes = {"conv": [("conv C->320 @64x64 (M 32768)", 1, 8, 64, 64, 320, (320, 640)), ("conv C->640 @32x32 (M 8192)", 1, 8, 32, 32, 640, (640, 1280)),
                   ("conv C->1280 @16x16 (M 2048)", 1, 8, 16, 16, 1280, (1280, 2560))],
          "dense": [("dense 8192x640xK", 0, 8, 32, 32, 640, (640, 2560)), ("dense 2048x1280xK", 0, 8, 16, 16, 1280, (1280, 5120)),
                    ("dense 32768x320xK", 0, 8, 64, 64, 320, (320, 1280))]}[kind]
for label, conv, B, H, W, N, cins in probes:
    for cfg in cfgs:
        us, kts = [], []
        for cin in cins:
            ms = C.c_float()
            r = lib().tsd_debug_gemm_bench(ctx.h, conv, B, H, W, cin, N, 1, 0, cfg, iters, C.byref(ms))
            if r != 0:
                us = None
                break
            us.append(ms.value * 1e3); kts.append((9 * cin if conv else cin) // 64)
        if not us:
            print(f"{label:34s} cfg {cfg:3d}: failed ({tsd._lib.lib().tsd_last_error().decode()[:80]})")
            continue
        slope = (us[1] - us[0]) / (kts[1] - kts[0]); fixed = us[0] - slope * kts[0]
        tf = [2.0 * B * H * W * N * 64 * kt / (u * 1e-6) / 1e12 for kt, u in zip(kts, us)]
        print(f"{label:34s} cfg {cfg:3d}: {kts[0]:3d} K tiles {us[0]:7.2f} us ({tf[0]:6.0f} TF)  {kts[1]:3d} K tiles {us[1]:7.2f} us ({tf[1]:6.0f} TF)  "
              f"slope {slope:6.3f} us/K-tile  fixed {fixed:6.2f} us")
