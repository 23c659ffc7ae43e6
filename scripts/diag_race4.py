"""Where does a rare run-to-run difference of generate() first appear?  The stress test's inputs (tests/test_gpu_models.py
test_generate_is_bitwise_repeatable_under_stress); N identical txt2img runs through the session API, the latents hashed after every
step and the images at the end.  Prints, for every run that differs from run 0, the first step whose latents differ."""
import os, sys, hashlib
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd import rng
from tsd.model import Session
SEED, B, L, T = 1234, 8, 64, 77
N = int(os.environ.get("N", 10)); STEPS = int(os.environ.get("STEPS", 50)); EVERY = int(os.environ.get("EVERY", 1))
d = tsd.Diffusion(seed=SEED); dec = tsd.Decoder(seed=SEED)
ctx = rng.normal(SEED, 771, B * T * 768).reshape(B, T, 768)
nl = B * 4 * L * L
lat0 = rng.normal(37, 2, nl).reshape(B, 4, L, L)
def h(a): return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
runs = []
for r in range(N):
    sess = Session(d.model, dec.model, B, L, T, cfg=False)
    sess.set_schedule(1000, STEPS, 0)
    n = sess.num_steps
    noise = rng.normal(37, 3, n * nl).reshape(n, B, 4, L, L)
    sess.upload(lat0, ctx, None, noise, 7.5)
    hs = []
    for i in range(n):
        sess.step(i)
        if i % EVERY == 0 or i == n - 1: hs.append((i, h(sess.latents())))
    sess.decode(); hs.append((-1, h(sess.images(rescale=True))))
    sess.close(); runs.append(hs)
bad = 0
for r in range(1, N):
    if runs[r] != runs[0]:
        bad += 1
        first = next(i for (i, a), (_, b) in zip(runs[0], runs[r]) if a != b)
        same_after = sum(1 for (i, a), (_, b) in zip(runs[0], runs[r]) if a == b)
        print(f"  run {r} differs from run 0: first at step {first} (-1 = only the decoded images); {same_after} of {len(runs[0])} checkpoints equal")
print(f"{N} runs, {bad} differ from run 0; final latents {sorted(set(x[-2][1] for x in runs))} images {sorted(set(x[-1][1] for x in runs))}")
