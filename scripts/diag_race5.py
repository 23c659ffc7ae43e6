"""Rate of run-to-run differences of the 50-step denoise loop on the stress test's inputs: N identical runs, final latents hashed."""
import os, sys, hashlib, collections, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd import rng
from tsd.model import Session
SEED, B, L, T = 1234, int(os.environ.get("B", 8)), 64, 77
N = int(os.environ.get("N", 200)); STEPS = int(os.environ.get("STEPS", 50))
d = tsd.Diffusion(seed=SEED)
ctx = rng.normal(SEED, 771, B * T * 768).reshape(B, T, 768)
nl = B * 4 * L * L
lat0 = rng.normal(37, 2, nl).reshape(B, 4, L, L)
sess = Session(d.model, None, B, L, T, cfg=False)
sess.set_schedule(1000, STEPS, 0)
n = sess.num_steps
noise = rng.normal(37, 3, n * nl).reshape(n, B, 4, L, L)
cnt = collections.Counter(); t0 = time.time()
for r in range(N):
    sess.upload(lat0, ctx, None, noise, 7.5)
    for i in range(n): sess.step(i)
    cnt[hashlib.sha1(sess.latents().tobytes()).hexdigest()[:8]] += 1
sess.close()
print(f"{N} x {n} steps in {time.time() - t0:.1f} s: {len(cnt)} distinct {dict(cnt)}")
