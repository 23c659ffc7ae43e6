"""C-ABI checks that need no GPU: the library loads, exports every symbol include/tsd.h declares, the
parameter inventory and FLOP census agree with the oracle/SURVEY, the RNG twins agree, and compute
entry points fail loudly (TSD_E_HIP) when no device is present - there is no CPU fallback."""
import ctypes as C

import numpy as np
import pytest

from oracle import rng as orng
from oracle import spec


def test_library_exports_every_declared_symbol(tsd_mod):
    lib = tsd_mod._lib.lib()
    names = tsd_mod._lib.declared_symbols()
    assert len(names) >= 50
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert lib.tsd_version() == 100


def test_param_inventory_matches_oracle_spec(tsd_mod):
    for kind, plist in (("diffusion", spec.diffusion_params()), ("decoder", spec.decoder_params()),
                        ("encoder", spec.encoder_params()), ("clip", spec.clip_params()),
                        ("diffusion_sd15", spec.diffusion_sd15_params()),
                        ("diffusion_sd15_torch", spec.diffusion_sd15_torch_params()),
                        ("clip_torch", spec.clip_torch_params()), ("decoder_torch", spec.decoder_torch_params()),
                        ("encoder_torch", spec.encoder_torch_params())):
        got = tsd_mod.param_specs(kind)
        assert len(got) == len(plist)
        for (name, shape, used, bound), p in zip(got, plist):
            assert name == p.name and tuple(shape) == tuple(p.shape) and used == p.used, name
            assert abs(bound - np.float32(p.bound)) <= 1e-9, name


def test_flop_census_matches_survey(tsd_mod):
    f = tsd_mod.flop_count
    assert abs(f("diffusion", 64) - 408.33) < 0.01 and abs(f("diffusion", 32) - 89.51) < 0.01
    assert abs(f("decoder", 64) - 2514.52) < 0.01 and abs(f("decoder", 32) - 622.19) < 0.01
    assert abs(f("encoder", 512) - 1116.66) < 0.01


def test_full_size_unet_inventory(tsd_mod):
    """BASELINE configs[4]: the full-size graph carries the 860 M parameters of an SD-1.5 UNet (859.3 M once the
    per-channel norm affines the reference's GroupNorm / LayerNorm do not have are left out) and twice the work."""
    plist = spec.diffusion_sd15_params()
    used = sum(p.numel for p in plist if p.used)
    assert abs(used / 1e6 - 859.32) < 0.01
    kinds = [k for k, _, _ in spec.FULL_UNET_STEPS]
    assert kinds.count("res") == 22 and kinds.count("attn") == 16 and kinds.count("upconv") == 3
    assert [f for _, _, f in spec.FULL_UNET_STEPS].count("push") == [f for _, _, f in spec.FULL_UNET_STEPS].count("pop") == 12
    assert abs(tsd_mod.flop_count("diffusion_sd15", 64) - 803.27) < 0.01


def test_rng_twins_agree():
    """The host mirror's counter RNG == the oracle's (the two files are written twice on purpose),
    both pinned by known answers so the pair cannot drift together; the DEVICE generator is pinned against the host one by
    tests/test_gpu_models.py::test_device_rng_equals_host_rng."""
    import tsd.rng as prng
    np.testing.assert_array_equal(prng.uniform(5, 77, 1000, 0.3), orng.uniform(5, 77, 1000, 0.3))
    np.testing.assert_array_equal(prng.normal(5, 78, 1000), orng.normal(5, 78, 1000))
    kat_u = np.array([-0.17280828952789307, -0.03223922476172447, 0.08573245257139206, 0.2056889683008194], dtype=np.float32)
    kat_n = np.array([0.04646738991141319, 0.09456849843263626, 1.1540602445602417, 0.024450836703181267], dtype=np.float32)
    np.testing.assert_array_equal(prng.uniform(5, 77, 4, 0.3), kat_u)
    np.testing.assert_array_equal(orng.normal(5, 78, 4), kat_n)
    u = prng.uniform(1, 2, 200000, 1.0)
    assert abs(u.mean()) < 0.01 and abs(u.std() - 1 / np.sqrt(3)) < 0.01 and u.min() >= -1 and u.max() < 1
    z = prng.normal(1, 3, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


def test_argument_errors_without_gpu(tsd_mod):
    lib = tsd_mod._lib.lib()
    assert lib.tsd_model_param_count(99) == tsd_mod._lib.TSD_E_ARG
    assert lib.tsd_ctx_synchronize(None) == tsd_mod._lib.TSD_E_ARG
    assert b"NULL" in lib.tsd_last_error()
    assert lib.tsd_flop_count(99, 64, 77) < 0


def test_no_cpu_fallback(tsd_mod):
    """Without a GPU every compute path must fail loudly, never silently run on the host."""
    lib = tsd_mod._lib.lib()
    if lib.tsd_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib.tsd_ctx_create(0, C.byref(h)) == tsd_mod._lib.TSD_E_HIP
    assert b"no CPU fallback" in lib.tsd_last_error()
    with pytest.raises(tsd_mod.TsdError):
        tsd_mod.Conv2D(4, 4, 3).forward(np.zeros((4, 8, 8), np.float32))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under the package or bench's product path may import it."""
    import os, re
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stable-diffusion.mojo_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, re.M), os.path.join(root, f)
