"""Real-checkpoint import (extension): safetensors parsing pinned against the `safetensors`
package, the diffusers SD-1.x UNet key map pinned by the published tensor / parameter counts of that model, and the
map's round trip.  No GPU needed (the GPU side is tests/test_gpu_models.py::test_checkpoint_import_*)."""
import numpy as np
import pytest

from oracle import spec
from util import randn


def test_safetensors_reader_and_writer_agree_with_the_safetensors_package(tsd_mod, tmp_path):
    st = pytest.importorskip("safetensors.numpy")
    from tsd import checkpoint as ck
    t = {"a.weight": randn(1, 3, 5), "b": randn(2, 7), "scalar_like": randn(3, 1)}
    for dtype in ("F32", "F16"):
        p = str(tmp_path / f"mine_{dtype}.safetensors")
        ck.write_safetensors(p, t, dtype)
        theirs = st.load_file(p)  # the package reads what this writer wrote
        mine = ck.read_safetensors(p)
        for k in t:
            np.testing.assert_array_equal(mine[k], theirs[k].astype(np.float32))
            np.testing.assert_allclose(mine[k], t[k], rtol=0 if dtype == "F32" else 1e-3, atol=0 if dtype == "F32" else 1e-3)
    q = str(tmp_path / "theirs.safetensors")
    st.save_file({k: v.astype(np.float16) for k, v in t.items()}, q)  # and this reader reads what the package wrote
    mine = ck.read_safetensors(q)
    for k in t:
        np.testing.assert_array_equal(mine[k], t[k].astype(np.float16).astype(np.float32))


def test_bf16_tensors_round_to_nearest_even(tsd_mod, tmp_path):
    torch = pytest.importorskip("torch")
    from tsd import checkpoint as ck
    x = randn(4, 1000) * 3.0
    p = str(tmp_path / "bf16.safetensors")
    ck.write_safetensors(p, {"x": x}, "BF16")
    want = torch.from_numpy(x).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(ck.read_safetensors(p)["x"], want)


def test_malformed_files_fail_loudly(tsd_mod, tmp_path):
    from tsd import checkpoint as ck
    p = str(tmp_path / "x.safetensors")
    ck.write_safetensors(p, {"x": randn(5, 4)})
    raw = bytearray(open(p, "rb").read())
    open(p, "wb").write(raw[:-4])  # truncated payload
    with pytest.raises(ValueError):
        ck.read_safetensors(p)


def test_key_map_matches_the_published_sd15_unet(tsd_mod):
    """An SD-1.x UNet checkpoint in diffusers layout holds 686 tensors / 859,520,964 parameters; the map must produce
    exactly that from the model's own inventory, cover every parameter the graph reads, and round-trip."""
    from tsd import checkpoint as ck
    plist = spec.diffusion_sd15_torch_params()
    assert sum(p.numel for p in plist if p.used) == 859_520_964
    P = {p.name: np.broadcast_to(np.float32(i + 1), p.shape) for i, p in enumerate(plist)}  # no 3.4 GB of data needed
    sd = ck.params_to_diffusers_sd15_unet(P)
    assert len(sd) == 686 and sum(int(np.prod(v.shape)) for v in sd.values()) == 859_520_964
    assert sd["down_blocks.1.resnets.0.conv_shortcut.weight"].shape == (640, 320, 1, 1)
    assert "down_blocks.0.resnets.0.conv_shortcut.weight" not in sd
    assert sd["up_blocks.3.attentions.2.transformer_blocks.0.ff.net.0.proj.weight"].shape == (2560, 320)
    assert sd["mid_block.attentions.0.transformer_blocks.0.attn2.to_k.weight"].shape == (1280, 768)
    back = ck.diffusers_sd15_unet_to_params(sd)
    assert {p.name for p in plist if p.used} <= set(back) <= {p.name for p in plist}
    for name, a in back.items():
        if "in_proj" not in name:
            assert np.array_equal(a, P[name]), name
    with pytest.raises(KeyError):
        ck.diffusers_sd15_unet_to_params({k: v for k, v in sd.items() if k != "conv_in.weight"})


def test_vae_key_map_matches_the_published_autoencoder(tsd_mod):
    """The SD-1.x VAE (diffusers AutoencoderKL) holds 248 tensors / 83,653,863 parameters: decoder + post_quant_conv
    49,490,199, encoder + quant_conv 34,163,664.  The maps must give exactly that and round-trip; older checkpoints name
    the attention projections query / key / value / proj_attn (some as 1x1 convolutions)."""
    from tsd import checkpoint as ck
    total = tensors = 0
    for which, plist, want in (("decoder", spec.decoder_torch_params(), 49_490_199), ("encoder", spec.encoder_torch_params(), 34_163_664)):
        assert sum(p.numel for p in plist if p.used) == want
        P = {p.name: np.broadcast_to(np.float32(i + 1), p.shape) for i, p in enumerate(plist)}
        sd = ck.params_to_diffusers_vae(P, which)
        assert sum(int(np.prod(v.shape)) for v in sd.values()) == want
        back = ck.diffusers_vae_to_params(sd, which)
        assert {p.name for p in plist if p.used} <= set(back) <= {p.name for p in plist}
        for name, a in back.items():
            if "in_proj" not in name:
                assert np.array_equal(a, P[name]), name
        mod = (ck.VAE_DECODER_MODULES[4] if which == "decoder" else ck.VAE_ENCODER_MODULES[14])
        old = dict(sd)
        for new, legacy in (("to_q", "query"), ("to_k", "key"), ("to_v", "value"), ("to_out.0", "proj_attn")):
            old[f"{mod}.{legacy}.weight"] = np.asarray(old.pop(f"{mod}.{new}.weight"))[:, :, None, None]
            old[f"{mod}.{legacy}.bias"] = old.pop(f"{mod}.{new}.bias")
        legacy_back = ck.diffusers_vae_to_params(old, which)
        for name in back:
            assert np.array_equal(legacy_back[name], back[name]), name
        total += want
        tensors += len(sd)
    assert total == 83_653_863 and tensors == 248
