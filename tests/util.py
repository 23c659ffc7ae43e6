"""Shared helpers for the parity tests."""
import numpy as np

from oracle import rng


def randn(tag, *shape, seed=7):
    return rng.normal(seed, tag, int(np.prod(shape))).reshape(shape)


def uni(tag, bound, *shape, seed=7):
    return rng.uniform(seed, tag, int(np.prod(shape)), bound).reshape(shape)


def rel_l2(y, ref):
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    return float(np.linalg.norm(y - ref) / max(np.linalg.norm(ref), 1e-30))


def max_rel(y, ref):
    y, ref = np.asarray(y, np.float64), np.asarray(ref, np.float64)
    return float(np.abs(y - ref).max() / max(np.abs(ref).max(), 1e-30))


# Stated tolerances (fp16 storage / fp32 accumulate on the device vs the fp32 oracle), BASELINE.md section 4:
# Every bound is <= 4x the largest error measured for its class (profiles/r02_gpu_parity_v5.log, r03_gpu_parity_*.log): ops
# 3.1e-4 / 7.9e-4, blocks 4.5e-4 / 6.5e-4, whole models 2.7e-3 / 3.1e-3 - a 10x regression of any kernel fails its test.
TOL_OP = 1.2e-3        # one op: relative L2
TOL_OP_MAX = 2.5e-3    # one op: max abs error / max |ref|
TOL_BLOCK = 2e-3       # residual / attention block: relative L2
TOL_BLOCK_MAX = 3e-3   # residual / attention block: max abs error / max |ref|
TOL_MODEL = 1e-2       # full UNet / decoder / encoder forward: relative L2
TOL_MODEL_MAX = 1.2e-2 # full model: max abs error / max |ref|


def assert_close(y, ref, tol_l2, tol_max=None, what=""):
    assert y.shape == ref.shape, f"{what}: shape {y.shape} vs {ref.shape}"
    assert np.isfinite(y).all(), f"{what}: non-finite output"
    e2, em = rel_l2(y, ref), max_rel(y, ref)
    print(f"[parity] {what}: rel_l2={e2:.3e} max_rel={em:.3e}")
    assert e2 <= tol_l2, f"{what}: rel_l2 {e2:.3e} > {tol_l2:.1e}"
    if tol_max is not None:
        assert em <= tol_max, f"{what}: max_rel {em:.3e} > {tol_max:.1e}"
