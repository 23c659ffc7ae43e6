"""Phase timestamps of the fused attention-tail kernel (build with CHAIN_TS=1: -DTSD_CHAIN_TS)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
import tsd
from tsd import rng
from tsd._lib import lib
tsd.set_strict(True)
B, L, T = int(os.environ.get("B", 8)), 64, 77
ctx = tsd.default_context()
d = tsd.Diffusion(seed=1234)
lat = rng.normal(1, 1, B*4*L*L).reshape(B,4,L,L); cx = rng.normal(1, 2, B*T*768).reshape(B,T,768)
s = tsd.Session(d.model, None, B, L, T); s.set_schedule(1000, 50, 0); s.upload(lat, cx, None, None)
for _ in range(3): s.step(1)
ctx.synchronize()
n = 512 * 16
buf = (C.c_ulonglong * n)()
KIND = int(os.environ.get("KIND", 0))
assert lib().tsd_debug_chain_ts(buf, n, KIND) == 0
a = np.array(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
if KIND == 1:
    names = ["prologue + GN -> A", "conv_in gemm+tok store", "LN + drain", "q, k gemms", "q/k store + drain", "V^T gemm", "V^T transpose+store", "-", "final wait"]
else:
  names = ["prologue issue", "S1 gemm (sa_out)", "LN1", "S3 gemm (q)", "cross attention", "S5 gemm + LN2", "FFN (10 chunks)", "b2+T, S9 gemm", "out store+stats"]
nblk = B * 64
for lo, hi, tag in (((0, 256, "round 1 (blocks 0..255)"), (256, 512, "round 2 (blocks 256..511)")) if nblk > 256 else ((0, nblk, f"all {nblk} blocks"),)):
    blk = a[lo:hi]
    print(tag, "median ticks per phase (s_memtime)")
    for i, nm in enumerate(names):
        dlt = blk[:, i + 1] - blk[:, i]
        print(f"  {nm:22s} med {int(np.median(dlt)):8d}  min {int(dlt.min()):8d}  max {int(dlt.max()):8d}")
    print(f"  {'total':22s} med {int(np.median(blk[:, 9] - blk[:, 0])):8d}")
    # marks 10..13 sit inside the tail's cross attention (round 4): after the post-q barrier | staging issued, q stored | context landed | head 0 done
    for nm, i, j in (("xattn: staging + q", 10, 11), ("xattn: landing wait", 11, 12), ("xattn: head 0", 12, 13)):
        dlt = blk[:, j] - blk[:, i]
        print(f"  {nm:22s} med {int(np.median(dlt)):8d}  min {int(dlt.min()):8d}  max {int(dlt.max()):8d}")
dur = (a[:nblk, 15] - a[:nblk, 14]) * 0.01
print(f"per-block duration (realtime): median {np.median(dur):.1f} us; s_memtime ticks per us: {np.median((a[:nblk, 9] - a[:nblk, 0]) / dur):.0f}")
rt = a[:nblk, 15]
print("realtime (100 MHz) span of last launch: first..last block end", (rt.max() - rt.min()) * 0.01, "us")
