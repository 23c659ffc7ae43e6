"""GPU: the fp16 STORAGE range of the path, and what happens beyond it.

The reference computes in fp32 (helpers/utils.mojo:12-15) and cannot overflow at any magnitude this model produces; libtsd
stores activations as fp16 (|x| <= 65504, 11 significant bits) and accumulates in fp32.  Every other parity case keeps its
activations O(1) (weights U(+-1/sqrt(fan_in)), affine-free norms, N(0,1) inputs), so they never approach that range.  Here:

 * conv / linear outputs, the residual stream of a residual block and the GEGLU product of an attention block are driven to
   1e3, 1e4 and 3e4 - still finite in fp16 - and held to the SAME relative tolerances as the O(1) cases (fp16's relative
   precision does not depend on the magnitude);
 * heavy-tailed weights (a few output channels x30, as trained SD checkpoints have) go through the residual / attention
   blocks and the VAE's last-level residual block;
 * cases that MUST overflow have a defined outcome: the call returns TSD_E_NONFINITE (never silent inf / NaN), the count is
   readable through tsd_debug_nonfinite_count, and the context is usable afterwards.
Supported dynamic range: BASELINE.md section 4.
"""
import threading

import numpy as np
import pytest

from cases import CASES, _attn_params, _res_params, _w
from oracle import models, ops
from util import TOL_BLOCK, TOL_BLOCK_MAX, TOL_OP, TOL_OP_MAX, assert_close, randn

pytestmark = pytest.mark.gpu

FP16_MAX = 65504.0
TARGETS = (1e3, 1e4, 3e4)


@pytest.fixture(scope="module")
def diffusion(gpu_ctx, tsd_mod):
    return tsd_mod.Diffusion(seed=1234)  # device-side counter-RNG init (as tests/test_gpu_models.py)


@pytest.fixture(scope="module")
def decoder(gpu_ctx, tsd_mod):
    return tsd_mod.Decoder(seed=1234)


def _conv_at(tsd, target, C=64, O=64, H=16, k=3, split=10.0):
    """Conv2D whose largest |output| is `target`: the amplitude is split between the weights (x `split`) and the input so that both
    operands stay inside fp16 as well."""
    x, w, b = randn(400, C, H, H), _w(401, O, C, k, k) * split, randn(402, O) * 0.1
    y1 = ops.conv2d(x, w, b, padding=(k // 2, k // 2))
    s = np.float32(target / np.abs(y1 - b[:, None, None]).max())
    x = x * s
    assert np.abs(x).max() < FP16_MAX and np.abs(w).max() < FP16_MAX
    ref = ops.conv2d(x, w, b, padding=(k // 2, k // 2))
    c = tsd.Conv2D(C, O, k, (k // 2, k // 2))
    c.kernel, c.bias = w, b
    return c, x, ref


@pytest.mark.parametrize("target", TARGETS)
def test_conv_output_magnitude(gpu_ctx, tsd_mod, target):
    c, x, ref = _conv_at(tsd_mod, target)
    assert 0.9 * target < np.abs(ref).max() < 1.1 * target
    assert_close(np.asarray(c.forward(x)), ref, TOL_OP, TOL_OP_MAX, what=f"conv3x3 64->64, max |y| = {target:g}")


@pytest.mark.parametrize("target", TARGETS)
def test_linear_output_magnitude(gpu_ctx, tsd_mod, target):
    M, K, N = 96, 320, 960
    x, w = randn(410, M, K), _w(411, N, K) * 10.0
    s = np.float32(target / np.abs(ops.linear(x, w, None)).max())
    x = x * s
    ref = ops.linear(x, w, None)
    l = tsd_mod.Linear(K, N, use_bias=False)
    l.weight = w
    assert_close(np.asarray(l.forward(x)), ref, TOL_OP, TOL_OP_MAX, what=f"linear 320->960, max |y| = {target:g}")


@pytest.mark.parametrize("name", ["unet_res_320_320", "unet_res_320_640"])
@pytest.mark.parametrize("target", TARGETS)
def test_residual_stream_magnitude(gpu_ctx, tsd_mod, name, target):
    """`x + ...` / `conv1x1(x) + ...` (diffusion.mojo:70-72) with the block input at `target`: the normalised branch is O(1), the
    skip path carries the magnitude to the output."""
    c = CASES[name]
    i = c.build()
    i["x"] = i["x"] * np.float32(target / np.abs(i["x"]).max())
    ref = np.asarray(c.oracle(i), np.float32)
    assert np.abs(ref).max() > 0.3 * target
    assert_close(np.asarray(c.device(tsd_mod, i), np.float32), ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"{name}, max |x| = {target:g}")


def _geglu_amplified(nh, ne, H, gain):
    """Attention-block parameters whose GEGLU projection (layer8, diffusion.mojo:136-141) is amplified so that a * gelu(g) - the
    widest-ranging intermediate of the block, fp16 in the op-by-op graph and in the fused tail's LDS tile - grows with gain^2."""
    C = nh * ne
    P = _attn_params("a", C, 70)
    P["a.layer8.weight"] = P["a.layer8.weight"] * np.float32(gain)
    # keep the block output O(gain^2) / 100: the second projection is left alone
    return dict(x=randn(60, C, H, H), c=randn(61, 77, 768), P=P)


@pytest.mark.parametrize("name,nh,ne,H", [("unet_attn_8x40", 8, 40, 16), ("unet_attn_8x80", 8, 80, 8)])
@pytest.mark.parametrize("gain", [8.0, 25.0, 45.0])
def test_geglu_product_magnitude(gpu_ctx, tsd_mod, name, nh, ne, H, gain):
    """a, g ~ gain * N(0, 0.6): the product a * gelu(g) peaks near (4 sigma)^2 = 6 * gain^2 -> 4e2, 4e3, 1.2e4 (fused tail at C = 320,
    op-by-op graph at C = 640)."""
    i = _geglu_amplified(nh, ne, H, gain)
    ref = np.asarray(models.unet_attention_block(i["P"], "a", i["x"], i["c"], nh, ne), np.float32)
    y = np.asarray(CASES[name].device(tsd_mod, i), np.float32)
    assert_close(y, ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"{name}, GEGLU projection x{gain:g}")


def _heavy_tail(w, rows, factor=30.0):
    w = w.copy()
    w[rows] *= np.float32(factor)
    return w


@pytest.mark.parametrize("name,cin,cout", [("unet_res_320_640", 320, 640), ("unet_res_640_1280_16", 640, 1280)])
def test_heavy_tailed_weights_residual_block(gpu_ctx, tsd_mod, name, cin, cout):
    """A few output channels of both 3x3 convolutions and of the 1x1 skip x30 (outlier channels of trained checkpoints)."""
    c = CASES[name]
    i = c.build()
    P = dict(i["P"])
    rows = [3, 77, cout - 5]
    for k in ("r.layer2.kernel", "r.layer5.kernel", "r.layer6.kernel"):
        P[k] = _heavy_tail(P[k], rows)
    i["P"] = P
    ref = np.asarray(c.oracle(i), np.float32)
    assert np.abs(ref).max() > 20.0
    assert_close(np.asarray(c.device(tsd_mod, i), np.float32), ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"{name}, 3 channels x30")


@pytest.mark.parametrize("name,nh,ne", [("unet_attn_8x40", 8, 40), ("unet_attn_8x80", 8, 80), ("unet_attn_8x160", 8, 160)])
def test_heavy_tailed_weights_attention_block(gpu_ctx, tsd_mod, name, nh, ne):
    """conv_in, out_proj, the GEGLU projections and conv_out with a few rows x30: outlier channels in the token stream (every
    LayerNorm row is then dominated by them), in (a, g) and in the block output."""
    C = nh * ne
    c = CASES[name]
    i = c.build()
    P = dict(i["P"])
    rows = [1, C // 2 + 3, C - 2]
    for k in ("a.layer2.kernel", "a.layer4.out_proj.weight", "a.layer6.out_proj.weight", "a.layer9.weight", "a.layer10.kernel"):
        P[k] = _heavy_tail(P[k], rows)
    P["a.layer8.weight"] = _heavy_tail(P["a.layer8.weight"], [5, 4 * C + 5, 4 * C - 1, 8 * C - 1])  # an (a, g) pair and two singles
    i["P"] = P
    ref = np.asarray(c.oracle(i), np.float32)
    assert_close(np.asarray(c.device(tsd_mod, i), np.float32), ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"{name}, outlier rows x30")


def test_heavy_tailed_weights_vae_last_level(gpu_ctx, tsd_mod):
    """The decoder's last-level residual block (vae.mojo:213-216: 256 -> 128 with the 1x1 skip) with outlier channels and an input
    two orders of magnitude above the other cases."""
    c = CASES["vae_res_256_128"]
    i = c.build()
    i["x"] = i["x"] * np.float32(100.0)
    P = dict(i["P"])
    for k in ("v.conv1.kernel", "v.conv2.kernel", "v.res_conv_layer.kernel"):
        P[k] = _heavy_tail(P[k], [0, 64, 127])
    i["P"] = P
    ref = np.asarray(c.oracle(i), np.float32)
    assert np.abs(ref).max() > 1e3
    assert_close(np.asarray(c.device(tsd_mod, i), np.float32), ref, TOL_BLOCK, TOL_BLOCK_MAX, what="vae_res_256_128, x100 input, 3 channels x30")


# ---- beyond the range: loud, never silent ------------------------------------------------------------------------------------------
def test_conv_op_beyond_fp16_is_still_exact(gpu_ctx, tsd_mod):
    """The op-level Conv2D writes its result as fp32 straight from the accumulator (no fp16 store on the way out), so an output
    beyond the fp16 range is not an overflow at all: it is the reference's value, and nothing is reported."""
    from tsd._lib import lib
    L = lib()
    L.tsd_debug_nonfinite_count(gpu_ctx.h, 1)
    c, x, ref = _conv_at(tsd_mod, 2.0e5)
    assert np.isfinite(ref).all() and np.abs(ref).max() > FP16_MAX
    assert_close(np.asarray(c.forward(x)), ref, TOL_OP, TOL_OP_MAX, what="conv3x3 64->64, max |y| = 2e5 (fp32 output)")
    assert L.tsd_debug_nonfinite_count(gpu_ctx.h, 0) == 0


def test_block_overflow_is_reported_not_silent(gpu_ctx, tsd_mod):
    """A residual block whose 1x1 skip term (diffusion.mojo:70-72) leaves the fp16 range: the reference's fp32 result is finite,
    the block's fp16 output tensor cannot hold it -> TSD_E_NONFINITE from the synchronous call, never silent inf / NaN."""
    from tsd._lib import TSD_E_NONFINITE, lib
    L = lib()
    assert L.tsd_debug_nonfinite_count(gpu_ctx.h, 1) >= 0
    c = CASES["unet_res_320_640"]
    i = c.build()
    i["x"] = i["x"] * np.float32(3.0e4 / np.abs(i["x"]).max())      # representable input ...
    P = dict(i["P"])
    P["r.layer6.kernel"] = P["r.layer6.kernel"] * np.float32(20.0)  # ... times an amplifying skip convolution
    i["P"] = P
    ref = np.asarray(c.oracle(i), np.float32)
    assert np.isfinite(ref).all() and np.abs(ref).max() > FP16_MAX
    with pytest.raises(tsd_mod.TsdError) as e:
        c.device(tsd_mod, i)
    assert e.value.code == TSD_E_NONFINITE, e.value
    assert "non-finite" in str(e.value)
    assert L.tsd_debug_nonfinite_count(gpu_ctx.h, 0) == 0          # reported once, then cleared
    # the context is usable afterwards and a representable problem is clean
    c2, x2, ref2 = _conv_at(tsd_mod, 1e3)
    assert_close(np.asarray(c2.forward(x2)), ref2, TOL_OP, TOL_OP_MAX, what="conv after an overflow report")
    assert L.tsd_debug_nonfinite_count(gpu_ctx.h, 0) == 0


def test_nonfinite_count_is_readable_without_failing(gpu_ctx, tsd_mod):
    """tsd_debug_nonfinite_count reads the context's counter; an input that is not finite counts like an overflow."""
    from tsd._lib import TSD_E_NONFINITE, lib
    L = lib()
    L.tsd_debug_nonfinite_count(gpu_ctx.h, 1)
    x = randn(420, 4, 32)
    x[1, 7] = np.inf
    with pytest.raises(tsd_mod.TsdError) as e:
        tsd_mod.LayerNorm(32).forward(x)
    assert e.value.code == TSD_E_NONFINITE
    tsd_mod.LayerNorm(32).forward(randn(421, 4, 32))              # clean again


def test_module_forward_overflow_is_reported(gpu_ctx, tsd_mod, diffusion):
    """`Diffusion.forward` on latents far outside the model's input range (1e6: the boundary conversion to fp16 already
    overflows): TSD_E_NONFINITE from the synchronous module call, and from the session at its next download."""
    from tsd._lib import TSD_E_NONFINITE, lib
    lat = randn(430, 1, 4, 8, 8) * np.float32(1e6)
    ctx = randn(431, 1, 77, 768)
    temb = ops.time_embedding(500.0)[None]
    with pytest.raises(tsd_mod.TsdError) as e:
        diffusion.forward(lat, ctx, temb)
    assert e.value.code == TSD_E_NONFINITE
    ok = diffusion.forward(lat / np.float32(1e6), ctx, temb)
    assert np.isfinite(ok).all()
    sess = tsd_mod.Session(diffusion.model, None, 1, 8, 77, cfg=False)
    sess.set_schedule(1000, 2, 0)
    sess.upload(lat[0:1], ctx, None, None)
    sess.step(0)                                                     # asynchronous: nothing to report yet
    with pytest.raises(tsd_mod.TsdError) as e2:
        sess.latents()
    assert e2.value.code == TSD_E_NONFINITE
    # the report is STICKY for the session (the context's counter was cleared by the first report; the NaN latents are still there):
    # a retry of the download, the next step and a decode all fail until upload() replaces the state
    assert lib().tsd_debug_nonfinite_count(gpu_ctx.h, 0) == 0
    for again in (sess.latents, lambda: sess.step(1)):
        with pytest.raises(tsd_mod.TsdError) as e3:
            again()
        assert e3.value.code == TSD_E_NONFINITE
    sess.upload(lat[0:1] / np.float32(1e6), ctx, None, None)
    sess.step(0)
    assert np.isfinite(sess.latents()).all()
    sess.close()
    assert lib().tsd_debug_nonfinite_count(gpu_ctx.h, 1) >= 0


def test_session_downloads_need_defined_state_and_only_own_data_poisons(gpu_ctx, tsd_mod, diffusion, decoder):
    """a download of state that was never written is TSD_E_STATE, not a host scan of uninitialised device memory that may
    latch the poison; and inf / NaN produced by ANOTHER model of the same context (reported through the context-wide counter at this
    session's next synchronisation point) is returned once but does not poison this session, whose own latents are finite."""
    from tsd._lib import TSD_E_NONFINITE, TSD_E_STATE
    B, L, T = 1, 8, 77
    sess = tsd_mod.Session(diffusion.model, decoder.model, B, L, T, cfg=False)
    sess.set_schedule(1000, 2, 0)
    with pytest.raises(tsd_mod.TsdError) as e0:
        sess.latents()
    assert e0.value.code == TSD_E_STATE
    lat, ctx = randn(440, B, 4, L, L), randn(441, B, T, 768)
    sess.upload(lat, ctx, None, None)
    with pytest.raises(tsd_mod.TsdError) as e1:
        sess.images()
    assert e1.value.code == TSD_E_STATE          # no decode() yet
    sess.step(0)
    # another model on the same context overflows: the context-wide counter is non-zero at this session's next synchronisation point
    with pytest.raises(tsd_mod.TsdError):
        diffusion.forward(lat * np.float32(1e6), ctx, tsd_mod.get_time_embedding(500.0).reshape(1, 320))
    try:
        got = sess.latents()                     # either clean (the module call reported and cleared the counter) ...
    except tsd_mod.TsdError as ex:               # ... or the context's code, once
        assert ex.code == TSD_E_NONFINITE
        got = sess.latents()                     # not latched: this session's latents are finite
    assert np.isfinite(got).all()
    sess.step(1)
    sess.decode()
    assert np.isfinite(sess.images()).all()
    sess.close()


# ---- two contexts, two threads, different settings (SURVEY.md section 8b "Threading") -----------------------------------------
def test_two_contexts_with_different_settings_from_two_threads(gpu_ctx, tsd_mod):
    """No process-global state: context A runs the fused attention-block kernels and the fused q/k/v projection, context B (same
    device, its own stream and workspace) the op-by-op graph with separate V^T launches and the separate skip GEMM - concurrently,
    from two host threads, several rounds.  Each must reproduce, bit for bit, what it computes alone on its thread."""
    from tsd._lib import Context, lib
    L = lib()
    ctx_b = Context(gpu_ctx.device)
    try:
        assert L.tsd_debug_set_fused_attention(ctx_b.h, 0) == 1
        assert L.tsd_debug_set_qkv_fuse(ctx_b.h, 0) == 1
        assert L.tsd_debug_set_res_fuse_skip(ctx_b.h, 0) == 1
        assert L.tsd_debug_set_attn_diag(ctx_b.h, 0) == 1
        # A's settings are untouched by B's setters
        for fn in (L.tsd_debug_set_fused_attention, L.tsd_debug_set_qkv_fuse, L.tsd_debug_set_res_fuse_skip, L.tsd_debug_set_attn_diag):
            assert fn(gpu_ctx.h, 1) == 1

        ia, ir = CASES["unet_attn_8x40"].build(), CASES["unet_res_320_640"].build()

        def attn(ctx):
            a = tsd_mod.Unet_Attention_Block(8, 40, ctx=ctx)
            P = ia["P"]
            a.layer2.kernel, a.layer2.bias = P["a.layer2.kernel"], P["a.layer2.bias"]
            a.layer4.in_proj.weight = P["a.layer4.in_proj.weight"]
            a.layer4.out_proj.weight, a.layer4.out_proj.bias = P["a.layer4.out_proj.weight"], P["a.layer4.out_proj.bias"]
            a.layer6.q_proj.weight, a.layer6.k_proj.weight = P["a.layer6.q_proj.weight"], P["a.layer6.k_proj.weight"]
            a.layer6.v_proj.weight = P["a.layer6.v_proj.weight"]
            a.layer6.out_proj.weight, a.layer6.out_proj.bias = P["a.layer6.out_proj.weight"], P["a.layer6.out_proj.bias"]
            a.layer8.weight, a.layer8.bias = P["a.layer8.weight"], P["a.layer8.bias"]
            a.layer9.weight, a.layer9.bias = P["a.layer9.weight"], P["a.layer9.bias"]
            a.layer10.kernel, a.layer10.bias = P["a.layer10.kernel"], P["a.layer10.bias"]
            return np.asarray(a.forward(ia["x"], ia["c"]), np.float32)

        def res(ctx):
            r = tsd_mod.Unet_Residual_Block(320, 640, ctx=ctx)
            P = ir["P"]
            r.layer2.kernel, r.layer2.bias = P["r.layer2.kernel"], P["r.layer2.bias"]
            r.layer3.weight, r.layer3.bias = P["r.layer3.weight"], P["r.layer3.bias"]
            r.layer5.kernel, r.layer5.bias = P["r.layer5.kernel"], P["r.layer5.bias"]
            r.layer6.kernel, r.layer6.bias = P["r.layer6.kernel"], P["r.layer6.bias"]
            return np.asarray(r.forward(ir["x"], ir["time"]), np.float32)

        alone = {"A": (attn(gpu_ctx), res(gpu_ctx)), "B": (attn(ctx_b), res(ctx_b))}
        # the two settings really are different code paths (fp32 vs fp16 residual stream, fused vs separate skip term) ...
        assert not np.array_equal(alone["A"][0], alone["B"][0]) and not np.array_equal(alone["A"][1], alone["B"][1])
        # ... and both are the reference's result
        ref_a = np.asarray(CASES["unet_attn_8x40"].oracle(ia), np.float32)
        for k in ("A", "B"):
            assert_close(alone[k][0], ref_a, TOL_BLOCK, TOL_BLOCK_MAX, what=f"context {k}, attention block")

        errors, rounds = [], 6

        def worker(tag, ctx):
            try:
                for _ in range(rounds):
                    ya, yr = attn(ctx), res(ctx)
                    if not (np.array_equal(ya, alone[tag][0]) and np.array_equal(yr, alone[tag][1])):
                        errors.append(f"context {tag}: result differs from its single-thread result")
            except Exception as ex:  # noqa: BLE001
                errors.append(f"context {tag}: {ex!r}")

        ts = [threading.Thread(target=worker, args=("A", gpu_ctx)), threading.Thread(target=worker, args=("B", ctx_b))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errors, errors
    finally:
        ctx_b.close()


def test_session_rejects_option_change_after_upload(gpu_ctx, tsd_mod, diffusion):
    """A session sized its workspace for the settings active at upload(): a tsd_debug_set_* call on its context afterwards makes
    step() fail with TSD_E_STATE (instead of running a graph the workspace was not planned for) until upload() is called again."""
    from tsd._lib import TSD_E_STATE, lib
    L = lib()
    lat, ctx = randn(440, 1, 4, 8, 8), randn(441, 1, 77, 768)
    sess = tsd_mod.Session(diffusion.model, None, 1, 8, 77, cfg=False)
    sess.set_schedule(1000, 2, 0)
    sess.upload(lat, ctx, None, None)
    sess.step(0)
    prev = L.tsd_debug_set_fused_attention(gpu_ctx.h, 0)
    try:
        with pytest.raises(tsd_mod.TsdError) as e:
            sess.step(1)
        assert e.value.code == TSD_E_STATE
        sess.upload(lat, ctx, None, None)
        sess.step(0)
        assert np.isfinite(sess.latents()).all()
    finally:
        L.tsd_debug_set_fused_attention(gpu_ctx.h, prev)
        sess.close()
