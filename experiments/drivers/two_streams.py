"""Experiment: the batch of 8 as two concurrent batch-4 sessions on two contexts (= two HIP streams) of one GPU."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd

def make(ctx, B):
    unet = tsd.Diffusion(seed=1234, ctx=ctx)
    L, T, n = 64, 77, 50
    lat = tsd.rng.normal(1, 1, B * 4 * L * L).reshape(B, 4, L, L)
    cx = tsd.rng.normal(1, 2, B * T * 768).reshape(B, T, 768)
    nz = tsd.rng.normal(1, 3, n * B * 4 * L * L).reshape(n, B, 4, L, L)
    s = tsd.Session(unet.model, None, B, L, T, cfg=False)
    s.set_schedule(1000, n, 0)
    s.upload(lat, cx, None, nz)
    return unet, s

K = 20
for nctx, B in ((1, 8), (2, 4), (2, 8), (4, 2)):
    ctxs = [tsd.Context(0) for _ in range(nctx)]
    objs = [make(c, B) for c in ctxs]
    for i in range(3):
        for _, s in objs: s.step(i)
    for c in ctxs: c.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        for _, s in objs: s.step(3 + i)
    for c in ctxs: c.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nctx} stream(s) x batch {B}: {1e3 * dt / K:.3f} ms per round of {nctx * B} samples -> {nctx * B * K / dt / 8:.1f} batch-8-equivalent steps/s")
    for _, s in objs: s.close()
