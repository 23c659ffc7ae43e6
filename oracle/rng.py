"""Counter-based RNG used for synthetic weights and inputs (oracle-side copy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws weights from Mojo's stdlib
PRNG (`helpers/utils.mojo:1722-1724` conv U(+-1/sqrt(fan_in)); `:1940-1944` linear), which is
not reproducible outside Mojo, so weights are *inputs* to both the oracle and the GPU path
(SURVEY.md Appendix A rule 3).  This generator is stateless: value = f(seed, tensor_id, index),
so the device can regenerate the same tensors without shipping big fixtures.  The product has
its own independent copy (device kernel `k_fill_uniform` + `tsd/rng.py`); tests check the two
agree bit-for-bit.
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    """splitmix64 finaliser on a uint64 array."""
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M2
    z ^= z >> np.uint64(27)
    z *= _M3
    z ^= z >> np.uint64(31)
    return z


def hash_u64(seed, tensor_id, n, offset=0):
    """64-bit hash for indices offset..offset+n-1 of tensor `tensor_id` under `seed`."""
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * _M1 + np.uint64(tensor_id) * _M2
        idx = np.arange(offset, offset + n, dtype=np.uint64)
        return _mix(idx + base)


def uniform(seed, tensor_id, n, bound, offset=0):
    """U(-bound, bound) float32, bit-reproducible on host and device.

    u24 = top 24 bits; v = u24*2^-23 - 1 (exact in fp32); w = v*bound (one fp32 rounding).
    """
    u = (hash_u64(seed, tensor_id, n, offset) >> np.uint64(40)).astype(np.float32)
    v = u * np.float32(2.0 ** -23) - np.float32(1.0)
    return (v * np.float32(bound)).astype(np.float32)


def normal(seed, tensor_id, n):
    """N(0,1) float32 via Box-Muller in float64 (host-only: inputs, never regenerated on device)."""
    h = hash_u64(seed, tensor_id, 2 * n)
    u1 = ((h[:n] >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 ** -53)
    u2 = ((h[n:] >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 ** -53)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)
