import os, sys, ctypes as C
sys.path.insert(0, "stable-diffusion.mojo_amd")
import tsd
from tsd._lib import lib
ctx = tsd.default_context()
for ms in (1.0, 5.0, 20.0, 100.0):
    tf, ghz = C.c_float(), C.c_float()
    r = lib().tsd_debug_mfma_sustained(ctx.h, ms, C.byref(tf), C.byref(ghz))
    print(f"target {ms} ms: rc={r} {tf.value:.1f} TFLOP/s at {ghz.value:.3f} GHz")
