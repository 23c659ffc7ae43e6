"""Graph-level pin of the oracle: the numpy restatement (oracle/models.py) against an independently written
torch.nn.functional restatement of the same Mojo sources (tests/torch_restatement.py), both in float64, on the same
seeded weights and inputs at an 8x8 latent.  The reference itself ships no vectors and cannot run here, so
this does not lift parity above "partial" - it removes single-author transcription errors from the graph level (block
wiring, skip/concat widths, group counts, eps values, head split, GEGLU order) the way the op-level pins do for the ops."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch_restatement as second
from oracle import models, ops, rng, spec

SEED = 1234


def _f64(P):
    return {k: np.asarray(v, dtype=np.float64) for k, v in P.items()}


def test_diffusion_forward_two_restatements_agree():
    P = _f64(spec.init_params("diffusion", SEED, only_used=True))
    L, T = 8, 77
    lat = rng.normal(SEED, 900, 4 * L * L).reshape(4, L, L).astype(np.float64)
    ctx = rng.normal(SEED, 901, T * 768).reshape(T, 768).astype(np.float64)
    temb = ops.time_embedding(500.0, dtype=np.float64).astype(np.float64)
    a = models.diffusion(P, lat, ctx, temb)
    b = second.diffusion(P, lat, ctx, temb)
    assert a.dtype == np.float64 and a.shape == b.shape == (4, L, L)
    err = np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"[pin] Diffusion.forward L=8: numpy oracle vs torch restatement rel_l2 = {err:.3e}")
    assert err < 1e-6


def test_decoder_forward_two_restatements_agree():
    P = _f64(spec.init_params("decoder", SEED, only_used=True))
    L = 8
    lat = rng.normal(SEED, 910, 4 * L * L).reshape(4, L, L).astype(np.float64) * 0.18215
    a = models.decoder(P, lat)
    b = second.decoder(P, lat)
    assert a.shape == b.shape == (3, 8 * L, 8 * L)
    err = np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"[pin] Decoder.forward L=8: numpy oracle vs torch restatement rel_l2 = {err:.3e}")
    assert err < 1e-6
