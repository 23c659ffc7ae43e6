"""pytest configuration: `gpu` marker, import paths, shared fixtures.

`-m "not gpu"` runs here (no GPU): oracle pins, golden fixtures, host logic, C-ABI load/symbol checks.
`-m gpu` runs on an MI355X box: parity of the HIP path against the oracle through the C ABI.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-diffusion.mojo_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def tsd_mod():
    import tsd
    return tsd


@pytest.fixture(scope="session")
def gpu_ctx(tsd_mod):
    """Default context on cuda:0.  A missing library is a FAILURE (no silent fallback); no device = skip."""
    lib = tsd_mod._lib.lib()  # raises TsdError if libtsd.so is not built
    if lib.tsd_device_count() == 0:
        pytest.skip("no HIP device visible")
    tsd_mod.set_strict(True)
    return tsd_mod.default_context()


@pytest.fixture(scope="session")
def unet_params():
    """Synthetic Diffusion weights from the oracle RNG (seed 1234), ~300 M floats, generated once."""
    from oracle import spec
    return spec.init_params("diffusion", 1234, only_used=True)
