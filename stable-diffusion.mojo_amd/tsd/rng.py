"""Counter-based RNG of the product: value = f(seed, tensor_id, index).

Host (numpy) twin of the device kernel `k_fill_uniform` (csrc/kernels_elementwise.hip); both produce
bit-identical float32 values, so synthetic weights never have to be shipped: the reference
random-initialises every struct in `__init__` (helpers/utils.mojo:1716-1727, :1938-1945) from Mojo's
irreproducible stdlib PRNG, hence weights/noise are INPUTS here (SURVEY.md App.A rule 3).
"""
import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    z = z.astype(np.uint64, copy=True)
    z ^= z >> np.uint64(30)
    z *= _M2
    z ^= z >> np.uint64(27)
    z *= _M3
    z ^= z >> np.uint64(31)
    return z


def hash_u64(seed, tensor_id, n, offset=0):
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * _M1 + np.uint64(tensor_id) * _M2
        return _mix(np.arange(offset, offset + n, dtype=np.uint64) + base)


def uniform(seed, tensor_id, n, bound):
    """U(-bound, bound) float32: u24 * 2^-23 - 1 (exact) times bound (one rounding)."""
    u = (hash_u64(seed, tensor_id, n) >> np.uint64(40)).astype(np.float32)
    v = u * np.float32(2.0 ** -23) - np.float32(1.0)
    return (v * np.float32(bound)).astype(np.float32)


def normal(seed, tensor_id, n):
    """N(0,1) float32 (Box-Muller in float64) - host-side inputs only (latents, context, step noise)."""
    h = hash_u64(seed, tensor_id, 2 * n)
    u1 = ((h[:n] >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 ** -53)
    u2 = ((h[n:] >> np.uint64(11)).astype(np.float64) + 0.5) * (2.0 ** -53)
    return (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)).astype(np.float32)


_next_id = [1 << 20]


def fresh_id():
    """tensor ids for ad-hoc op-level structs (module-level models use kind*4096+index)."""
    _next_id[0] += 1
    return _next_id[0]
