"""Host-side mirror of `pipeline.generate` (pipeline.mojo:12-127): [tokenizer -> CLIP ->] sampler -> [encoder] ->
denoise loop (UNet x1 or x2 with CFG) -> decoder -> rescale.  The context embedding is normally an input (the
measured path starts there); `encode_prompts` is the reference's prompt front end (pipeline.mojo:31-54).  Batched
over independent prompts (the reference is batch 1; `pipeline.mojo:12` suggests exactly this batching)."""
import numpy as np

from . import rng
from .model import Session
from .utils import rescale


def encode_prompts(prompts, tokenizer, clip):
    """pipeline.mojo:39-54: ids = bpe_encode(prompt.replace(" ", "</w>")) -> CLIP.forward -> (B, 77, 768).
    Prompts longer than 77 ids are cut to 77 (the reference's `set_items` into a 77-wide row, clip.mojo:91-93)."""
    from .tokenizer import process_prompt
    if isinstance(prompts, str):
        prompts = [prompts]
    ids = np.zeros((len(prompts), 77), dtype=np.int32)
    for i, p in enumerate(prompts):
        t = tokenizer.bpe_encode(process_prompt(p))[:77]
        ids[i, : len(t)] = t
    return clip.forward(ids)


def generate(diffusion, decoder, context, uncond_context=None, strength=0.8, cfg=True, cfg_scale=7.5,
             inference_steps=50, seed_val=0, input_image=None, encoder=None, latents=None, noise=None,
             num_training_steps=1000, L=64, return_latents=False):
    """context (B,T,768); returns images (B,3,8L,8L) in [0,255] like pipeline.mojo:127.

    latents / noise default to N(0,1) from the counter RNG keyed by seed_val (App.A D19)."""
    context = np.asarray(context, dtype=np.float32)
    if context.ndim == 2:
        context = context[None]
    B, T, _ = context.shape
    if not (0.0 <= strength <= 1.0):  # pipeline.mojo:23-29
        print("Strength must be between 0 and 1. Returning empty matrix")
        return np.zeros((0, 0, 0), dtype=np.float32)
    sess = Session(diffusion.model, decoder.model if decoder is not None else None, B, L, T, cfg=cfg)
    start = 0
    if input_image is not None:
        start = inference_steps - int(inference_steps * strength)  # sampler.mojo:68-70
    sess.set_schedule(num_training_steps, inference_steps, start)
    n = sess.num_steps
    nl = B * 4 * L * L
    if input_image is not None:  # pipeline.mojo:66-79
        img = rescale(input_image, (0, 255), (-1, 1))
        enc_noise = rng.normal(seed_val, 1, nl).reshape(B, 4, L, L)
        latents = encoder.forward(img, enc_noise)
    elif latents is None:
        latents = rng.normal(seed_val, 2, nl).reshape(B, 4, L, L)
    if noise is None:
        noise = rng.normal(seed_val, 3, n * nl).reshape(n, B, 4, L, L)
    sess.upload(latents, context, uncond_context if cfg else None, noise, cfg_scale)
    if input_image is not None:
        sess.add_noise(0, rng.normal(seed_val, 4, nl).reshape(B, 4, L, L))  # sampler.mojo:111-124 at timesteps[0]
    for i in range(n):  # pipeline.mojo:87-122
        sess.step(i)
    out_lat = sess.latents()
    if decoder is None or return_latents:
        sess.close()
        return out_lat
    sess.decode()
    images = sess.images(rescale=True)
    sess.close()
    return images
