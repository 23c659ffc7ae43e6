"""Regenerates tests/golden/golden.npz: oracle outputs for the cases listed in tests/cases.py::GOLDEN.

Run from the repo root:  python tests/golden/make_golden.py
Inputs are NOT stored: every case rebuilds them from the seeded counter RNG (oracle/rng.py), so the
fixture holds only the expected outputs (float32).  The reference ships no fixtures of its own
(SURVEY.md section 4); these pin the oracle against regressions and give the GPU tests a target that
does not depend on the oracle code at run time.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cases import CASES, GOLDEN  # noqa: E402


MAX_ELEMS = 8192


def pack(y):
    """(values, meta): outputs larger than MAX_ELEMS are stored as a fixed-stride subsample plus float64
    checksums (sum, sum of squares) of the full tensor."""
    flat = np.asarray(y, dtype=np.float32).reshape(-1)
    stride = max(1, -(-flat.size // MAX_ELEMS))
    meta = np.array([flat.size, stride, flat.astype(np.float64).sum(), (flat.astype(np.float64) ** 2).sum()])
    return flat[::stride].copy(), meta


def main():
    out = {}
    for name in GOLDEN:
        c = CASES[name]
        y = np.asarray(c.oracle(c.build()), dtype=np.float32)
        out[name], out[name + "__meta"] = pack(y)
        print(f"{name:36s} {y.shape} -> {out[name].size} values")
    path = os.path.join(HERE, "golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1024, "KiB")


if __name__ == "__main__":
    main()
