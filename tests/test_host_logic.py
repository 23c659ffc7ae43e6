"""Host-side mirror logic (no GPU): the product's DDPMSampler mirror against the oracle's restatement of
sampler.mojo, and the reference-convention error handling of the Python structs."""
import numpy as np

from oracle import sampler as osampler


def test_sampler_mirror_matches_oracle(tsd_mod):
    a, b = tsd_mod.DDPMSampler(0, 1000), osampler.DDPMSampler(1000)
    np.testing.assert_array_equal(a.alphas_cumprod, b.alphas_cumprod)
    for n in (1, 10, 50):
        a.set_inference_timesteps(n)
        b.set_inference_timesteps(n)
        np.testing.assert_array_equal(a.timesteps, b.timesteps)
        for t in a.timesteps[:5]:
            assert a.get_previous_timestep(int(t)) == b.previous_timestep(int(t))
            assert a.get_variance(int(t)) == b.variance(int(t))
    a.set_strength(0.6)
    b.set_strength(0.6)
    np.testing.assert_array_equal(a.timesteps, b.timesteps)
    assert a.start_step == b.start_step == 20


def test_generate_rejects_bad_strength(tsd_mod, capsys):
    out = tsd_mod.generate(None, None, np.zeros((1, 77, 768), np.float32), strength=1.5)  # pipeline.mojo:23-29
    assert out.shape == (0, 0, 0) and "Strength must be between 0 and 1" in capsys.readouterr().out


def test_linear_rejects_bad_input_without_touching_gpu(tsd_mod, capsys):
    y = tsd_mod.Linear(16, 8).forward(np.zeros((1, 3, 17), np.float32))  # helpers/utils.mojo:1955-1957
    assert y.shape == (0, 0, 0) and "Returning null matrix" in capsys.readouterr().out
