"""Run-to-run determinism of the other paths: N identical calls each, distinct results counted.
   cfg: 20-step guided loop (UNet batch 16) ; decode / encode: the VAE halves on 8 x 512^2 ; sd15: full-size UNet forward, batch 4 ;
   clip: text encoder, 8 prompts ; b1 / b3: the headline loop at batch 1 / 3 (other tile counts, partial tiles)."""
import os, sys, hashlib, collections, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd import rng
from tsd.model import Session
SEED = 1234
def h(a): return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:8]
def report(name, cnt, t0): print(f"{name}: {sum(cnt.values())} calls in {time.time() - t0:.1f} s: {len(cnt)} distinct {dict(cnt)}", flush=True)
def loop(name, B, cfg, steps, N, d):
    L, T = 64, 77
    ctx = rng.normal(SEED, 771, B * T * 768).reshape(B, T, 768); unc = rng.normal(SEED, 772, B * T * 768).reshape(B, T, 768)
    nl = B * 4 * L * L; lat0 = rng.normal(37, 2, nl).reshape(B, 4, L, L)
    sess = Session(d.model, None, B, L, T, cfg=cfg); sess.set_schedule(1000, steps, 0); n = sess.num_steps
    noise = rng.normal(37, 3, n * nl).reshape(n, B, 4, L, L)
    cnt = collections.Counter(); t0 = time.time()
    for r in range(N):
        sess.upload(lat0, ctx, unc if cfg else None, noise, 7.5)
        for i in range(n): sess.step(i)
        cnt[h(sess.latents())] += 1
    sess.close(); report(name, cnt, t0)
modes = (sys.argv[1] if len(sys.argv) > 1 else "cfg,b1,b3,decode,encode,sd15,clip").split(",")
N = int(os.environ.get("N", 100))
d = tsd.Diffusion(seed=SEED) if any(m in modes for m in ("cfg", "b1", "b3")) else None
if "cfg" in modes: loop("cfg (UNet batch 16, 20 steps)", 8, True, 20, N, d)
if "b1" in modes: loop("batch 1, 50 steps", 1, False, 50, N, d)
if "b3" in modes: loop("batch 3, 50 steps", 3, False, 50, N, d)
if "decode" in modes:
    dec = tsd.Decoder(seed=SEED); lat = rng.normal(SEED, 5, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)
    cnt = collections.Counter(); t0 = time.time()
    for r in range(N): cnt[h(dec.forward(lat))] += 1
    report("decode 8 x 512^2", cnt, t0); dec.model.close()
if "encode" in modes:
    enc = tsd.Encoder(seed=SEED); img = rng.uniform(SEED, 6, 8 * 3 * 512 * 512, 1.0).reshape(8, 3, 512, 512); nz = rng.normal(SEED, 7, 8 * 4 * 64 * 64).reshape(8, 4, 64, 64)
    cnt = collections.Counter(); t0 = time.time()
    for r in range(N): cnt[h(enc.forward(img, nz))] += 1
    report("encode 8 x 512^2", cnt, t0); enc.model.close()
if "sd15" in modes:
    if d is not None: d.model.close()
    f = tsd.Diffusion(seed=SEED, variant="diffusion_sd15"); B = 4
    lat = rng.normal(SEED, 8, B * 4 * 64 * 64).reshape(B, 4, 64, 64); ctx = rng.normal(SEED, 9, B * 77 * 768).reshape(B, 77, 768)
    te = np.stack([tsd.get_time_embedding(500.0)] * B) if hasattr(tsd, "get_time_embedding") else rng.normal(SEED, 10, B * 320).reshape(B, 320)
    cnt = collections.Counter(); t0 = time.time()
    for r in range(2 * N): cnt[h(f.forward(lat, ctx, te))] += 1
    report("full-size UNet forward, batch 4", cnt, t0); f.model.close()
if "clip" in modes:
    c = tsd.CLIP(seed=SEED); tok = (np.arange(8 * 77).reshape(8, 77) * 37 % 49000).astype(np.int32)
    cnt = collections.Counter(); t0 = time.time()
    for r in range(2 * N): cnt[h(c.forward(tok))] += 1
    report("CLIP text encoder, 8 prompts", cnt, t0)
