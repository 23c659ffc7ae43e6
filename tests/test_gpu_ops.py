"""GPU parity: every op- and block-level C-ABI entry point against the CPU oracle on the same seeded inputs.

Tolerances are stated in tests/util.py (fp16 storage / fp32 accumulate on the device vs fp32 oracle)."""
import numpy as np
import pytest

from cases import CASES
from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", sorted(CASES))
def test_case_matches_oracle(name, gpu_ctx, tsd_mod):
    c = CASES[name]
    inputs = c.build()
    ref = np.asarray(c.oracle(inputs), dtype=np.float32)
    y = np.asarray(c.device(tsd_mod, inputs), dtype=np.float32)
    if c.tol == 0.0:
        np.testing.assert_array_equal(y, ref)  # pure data movement: bit-exact
    else:
        assert_close(y, ref, c.tol, c.tol_max, what=name)


def test_inputs_not_mutated(gpu_ctx, tsd_mod):
    """The reference mutates inputs through aliasing (App.A D14/D16); the C ABI must not."""
    c = CASES["unet_res_320_320"]
    i = c.build()
    x0, t0 = i["x"].copy(), i["time"].copy()
    c.device(tsd_mod, i)
    np.testing.assert_array_equal(i["x"], x0)
    np.testing.assert_array_equal(i["time"], t0)


def test_shape_errors_follow_reference_convention(gpu_ctx, tsd_mod, capsys):
    """Shape errors: status TSD_E_SHAPE -> print + null matrix (helpers/utils.mojo:1955-1957) unless strict."""
    tsd_mod.set_strict(False)
    try:
        y = tsd_mod.GroupNorm(32, 64).forward(np.zeros((32, 4, 4), np.float32))  # num_channels > C (:1847-1849)
        assert y.shape == (0, 0, 0)
        assert "null matrix" in capsys.readouterr().out
        y = tsd_mod.Linear(16, 8).forward(np.zeros((1, 3, 17), np.float32))
        assert y.shape == (0, 0, 0)
    finally:
        tsd_mod.set_strict(True)
    with pytest.raises(tsd_mod.TsdError):
        tsd_mod.GroupNorm(32, 64).forward(np.zeros((32, 4, 4), np.float32))


def test_upsample_fusion_equals_materialised(gpu_ctx, tsd_mod):
    """Upsample folded into the next conv's addressing == explicit Upsample then Conv2D (op level)."""
    from oracle import ops
    from util import randn, uni
    x = randn(300, 64, 6, 6)
    w = uni(301, 0.05, 64, 64, 3, 3)
    c = tsd_mod.Conv2D(64, 64, 3, (1, 1))
    c.kernel = w
    y = c.forward(tsd_mod.Upsample(2).forward(x))
    ref = ops.conv2d(ops.upsample_nearest2x(x), w, None, padding=(1, 1))
    assert_close(y, ref, 3e-3, 1e-2, "upsample+conv")


@pytest.mark.parametrize("conv,B,H,Cin,N,stride,ups", [
    (1, 2, 16, 320, 320, 1, 0), (1, 8, 32, 640, 640, 1, 0), (1, 2, 16, 320, 320, 2, 0), (1, 2, 8, 640, 320, 1, 1),
    (0, 8, 64, 320, 320, 1, 0), (0, 8, 16, 1280, 1280, 1, 0), (0, 1, 8, 320, 160, 1, 0), (0, 3, 8, 64, 320, 1, 0),
    (0, 2, 16, 256, 512, 1, 0), (1, 1, 32, 128, 128, 1, 0)])
def test_tile_configurations_are_bitwise_equivalent(gpu_ctx, tsd_mod, conv, B, H, Cin, N, stride, ups):
    """Every production tile configuration of the GEMM / conv kernel gives bit-identical results (each output element
    accumulates its K products in the same order whatever the tile, ring depth or wave count) - the property behind
    the batch invariance of the whole path."""
    import ctypes as C
    from tsd._lib import lib
    d, m = C.c_float(), C.c_float()
    ref = 1 if N % 160 == 0 else 3
    # 45-53: the same tiles with four loader waves issuing the block's LDS-DMA stream (round 3; opt-in)
    for cfg in (0, 1, 2, 3, 5, 6, 7, 8, 9, 10, 11, 13, 45, 46, 47, 48, 49, 50, 51, 53, 54, 55):
        if N % 160 and cfg in (0, 1, 5, 6, 7, 11, 45, 46, 47, 51, 54):
            continue
        r = lib().tsd_debug_gemm_check(gpu_ctx.h, conv, B, H, H, Cin, N, stride, ups, cfg, ref, C.byref(d), C.byref(m))
        assert r == 0, f"cfg {cfg}: rc {r}"
        assert d.value == 0.0 and m.value > 0.0, f"cfg {cfg} differs from cfg {ref}: max|diff| {d.value} (max|ref| {m.value})"


@pytest.mark.parametrize("B,H,Cin", [(2, 16, 320), (1, 24, 128), (8, 64, 320)])
def test_thin_tile_configurations_are_bitwise_equivalent(gpu_ctx, B, H, Cin):
    """N <= 16 convolutions (the UNet's 320 -> 4 and the decoder's 128 -> 3 output layers): the 64-row / 4-slot tile the dispatcher
    picks for few-tile problems against the 128-row / 2-slot one - same K order, identical bits."""
    import ctypes as C
    from tsd._lib import lib
    d, m = C.c_float(), C.c_float()
    r = lib().tsd_debug_gemm_check(gpu_ctx.h, 1, B, H, H, Cin, 4, 1, 0, 24, 4, C.byref(d), C.byref(m))
    assert r == 0, r
    assert d.value == 0.0 and m.value > 0.0, (d.value, m.value)


def test_attention_exact_pass_runs_only_when_a_row_overflows(gpu_ctx, tsd_mod):
    """kernels_attn.hip runs the softmax optimistically (reference = first key tile's row maximum + 4, in log2 units) and
    repeats a workgroup with the exact running maximum only if an fp16 probability overflowed.  Ordinary inputs must never
    take the second pass; scores that climb by tens of log2 units along the keys must (and still match the oracle)."""
    from tsd._lib import lib
    L = lib()
    assert L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1) >= 0
    for name, expect_exact in (("self_attention_d40", False), ("self_attention_d40_falling_scores", False),
                               ("self_attention_d40_rising_scores", True), ("self_attention_d80_rising_scores", True),
                               ("self_attention_d160_rising_scores", True),
                               # key loops of 17 - 18 tiles: the early look (every eighth tile) and the final check both lead to the exact pass
                               ("self_attention_d40_long_rising", True), ("self_attention_d40_spike_tile14", True),
                               ("self_attention_d40_spike_last_tile", True), ("self_attention_d80_long_rising", True)):
        c = CASES[name]
        i = c.build()
        y = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
        n = L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
        assert (n > 0) == expect_exact, (name, n)
        assert_close(y, np.asarray(c.oracle(i), dtype=np.float32), c.tol, c.tol_max, what=name)


def test_attention_second_reference_keeps_self_peaked_rows_in_the_optimistic_pass(gpu_ctx, tsd_mod):
    """Heavy-tailed scores as trained SD self-attention produces them (a token's own key 20+ log2 units above everything in key
    tile 0): with the tile-0-only reference of round 2 nearly every workgroup repeats exactly (2x on the largest kernel of
    the step); with the second reference - the row maximum over the query's own 32-key block - none does.  The repeat RATE is
    asserted, and both paths match the oracle."""
    from tsd._lib import lib
    L = lib()
    c = CASES["self_attention_d40_self_peaked"]
    i = c.build()
    ref = np.asarray(c.oracle(i), dtype=np.float32)
    n_wg = 8 * ((320 + 127) // 128)  # heads x 128-query workgroups (S = 320 runs the 32-queries-per-wave variant)
    prev = L.tsd_debug_set_attn_diag(gpu_ctx.h, 0)
    try:
        L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
        y0 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
        n0 = L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
        L.tsd_debug_set_attn_diag(gpu_ctx.h, 1)
        y1 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
        n1 = L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
    finally:
        L.tsd_debug_set_attn_diag(gpu_ctx.h, prev)
    print(f"[parity] self-peaked scores: exact repeats {n0}/{n_wg} workgroups with the tile-0 reference, {n1}/{n_wg} with the own-block reference")
    assert n0 >= n_wg // 2, n0          # the round-2 reference repeats (at least the workgroups beyond key tile 0)
    assert n1 == 0, n1                  # the second reference keeps every row inside fp16
    assert_close(y0, ref, c.tol, c.tol_max, what="self-peaked scores, tile-0 reference (exact repeat)")
    assert_close(y1, ref, c.tol, c.tol_max, what="self-peaked scores, own-block reference (optimistic pass)")


@pytest.mark.parametrize("name", ["unet_res_320_640", "unet_res_640_1280_16", "unet_res_2560_1280_8", "unet_res_ragged_128_192", "vae_res_256_128"])
def test_fused_skip_conv_equals_separate_skip_gemm(gpu_ctx, tsd_mod, name):
    """`conv1x1(x) + conv3x3(h)` of a residual block (diffusion.mojo:70-72, vae.mojo:65-67): the 1x1 convolution as extra K of the
    3x3 one (default) against the separate GEMM + residual add.  Not bitwise - the fused form keeps the skip term in the fp32
    accumulator instead of rounding it to fp16 first - so both are held to the oracle and to each other."""
    from tsd._lib import lib
    from util import rel_l2
    L = lib()
    c = CASES[name]
    i = c.build()
    ref = np.asarray(c.oracle(i), dtype=np.float32)
    prev = L.tsd_debug_set_res_fuse_skip(gpu_ctx.h, 0)
    try:
        y0 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
        L.tsd_debug_set_res_fuse_skip(gpu_ctx.h, 1)
        y1 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
    finally:
        L.tsd_debug_set_res_fuse_skip(gpu_ctx.h, prev)
    assert_close(y0, ref, c.tol, c.tol_max, what=f"{name}, separate skip GEMM")
    assert_close(y1, ref, c.tol, c.tol_max, what=f"{name}, skip fused into conv2")
    d = rel_l2(y1, y0)
    print(f"[parity] {name}: fused vs separate skip rel_l2={d:.3e}")
    assert 0.0 < d <= c.tol, d  # two different paths (d > 0) that agree


@pytest.mark.parametrize("name", ["self_attention_d40", "self_attention_d80", "self_attention_d160", "self_attention_vae_1head",
                                  "self_attention_d80_T96", "self_attention_d80_T96_bias",
                                  "unet_attn_8x80", "unet_attn_8x160", "vae_attention_512"])
def test_fused_qkv_projection_equals_two_launches(gpu_ctx, tsd_mod, name):
    """helpers/attention.mojo:29-31 `in_proj` + chunk: q | k token-major and V^T channel-major from ONE GEMM whose tiles beyond column
    2C store transposed (default) against the q/k GEMM + the swapped-operand V^T GEMM.  Same products in the same K order - the
    outputs are expected to agree bit for bit; both are held to the oracle."""
    from tsd._lib import lib
    from util import rel_l2
    L = lib()
    c = CASES[name]
    i = c.build()
    ref = np.asarray(c.oracle(i), dtype=np.float32)
    prev = L.tsd_debug_set_qkv_fuse(gpu_ctx.h, 0)
    try:
        y0 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
        L.tsd_debug_set_qkv_fuse(gpu_ctx.h, 1)
        y1 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
    finally:
        L.tsd_debug_set_qkv_fuse(gpu_ctx.h, prev)
    assert_close(y0, ref, c.tol, c.tol_max, what=f"{name}, two launches")
    assert_close(y1, ref, c.tol, c.tol_max, what=f"{name}, fused q/k/v projection")
    d = rel_l2(y1, y0)
    print(f"[parity] {name}: fused vs separate q/k/v projection rel_l2={d:.3e}")
    assert d <= 1e-4, d


def test_mfma_sustained_probe_reports_a_plausible_ceiling(gpu_ctx):
    """tsd_debug_mfma_sustained: register-resident fp16 MFMA loop; the figure bench.py prints next to the nominal 2.5 PF."""
    import ctypes as C
    from tsd._lib import lib
    tf, ghz = C.c_float(), C.c_float()
    assert lib().tsd_debug_mfma_sustained(gpu_ctx.h, 5.0, C.byref(tf), C.byref(ghz)) == 0
    assert 500.0 < tf.value < 2600.0, tf.value       # cannot exceed the nominal dense peak
    assert 0.8 < ghz.value < 2.6, ghz.value
    # cycles: 1024 SIMDs x 1024 flop/clk at that clock bounds it from above (block 0's clock sample, so allow 15 %)
    assert tf.value <= 1.15 * 1024 * 1024 * ghz.value * 1e9 / 1e12
    assert lib().tsd_debug_mfma_sustained(gpu_ctx.h, -1.0, C.byref(tf), C.byref(ghz)) != 0


def test_attention_64_queries_per_wave_is_bitwise_the_128_query_workgroup(gpu_ctx, tsd_mod):
    """flash_attn_kernel<40, 2> (64 queries per wave, picked for long key loops on big grids) against <40, 1>: in the
    optimistic pass every row goes through the same instruction sequence, so the outputs must be identical bit for bit,
    ragged lengths included.  In the exact repeat the decision to move the softmax reference is taken per wave (any of
    its rows), so there the two only agree to rounding; both must match the oracle."""
    from tsd._lib import lib
    L = lib()
    prev = L.tsd_debug_set_attn_qb(gpu_ctx.h, 1)
    L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
    try:
        for name in ("self_attention_d40", "self_attention_d40_ragged", "self_attention_d40_rising_scores",
                     "self_attention_d40_falling_scores", "cross_attention_d40_T77"):
            c = CASES[name]
            i = c.build()
            L.tsd_debug_set_attn_qb(gpu_ctx.h, 1)
            y1 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            L.tsd_debug_set_attn_qb(gpu_ctx.h, 2)
            y2 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            if "rising" in name:
                assert L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1) > 0
                assert_close(y2, y1, 1e-3, 1e-2, what=name + " (QB=2 vs QB=1, exact repeat)")  # fp16 P rounds differently once the reference moves
            else:
                assert L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1) == 0
                np.testing.assert_array_equal(y1, y2, err_msg=name)
            assert_close(y2, np.asarray(c.oracle(i), dtype=np.float32), c.tol, c.tol_max, what=name + " (QB=2)")
    finally:
        L.tsd_debug_set_attn_qb(gpu_ctx.h, prev)


def test_attention_8wave_two_group_kernel_matches_the_oracle_and_the_4wave_kernel(gpu_ctx, tsd_mod):
    """flash_attn8_kernel<40> (kernels_attn8.hip: eight waves, two groups one phase apart, 512 queries per workgroup - what the 4096 x 4096
    call of the 64x64 level runs) forced onto every d = 40 case, ragged lengths, partial last tiles, key loops of 1 - 18 tiles, the
    cases that must take the exact repeat (early look and final check) and the 77-key cross attention included.  Same products in the
    same order as flash_attn_kernel<40, 2>, but the softmax reference enters through the QK^T pad column (rounded to 32 x fp16), so the two
    agree to the fp16 rounding of P rather than bit for bit; each is held to the oracle with the case's own tolerance."""
    from tsd._lib import lib
    from util import rel_l2
    L = lib()
    prev = L.tsd_debug_set_attn_qb(gpu_ctx.h, 2)
    L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
    try:
        for name, expect_exact in (("self_attention_d40", False), ("self_attention_d40_ragged", False), ("self_attention_d40_falling_scores", False),
                                   ("self_attention_d40_self_peaked", False), ("cross_attention_d40_T77", False),
                                   ("self_attention_d40_rising_scores", True), ("self_attention_d40_long_rising", True),
                                   ("self_attention_d40_spike_tile14", True), ("self_attention_d40_spike_last_tile", True)):
            c = CASES[name]
            i = c.build()
            ref = np.asarray(c.oracle(i), dtype=np.float32)
            L.tsd_debug_set_attn_qb(gpu_ctx.h, 2)
            y2 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
            L.tsd_debug_set_attn_qb(gpu_ctx.h, 3)
            y8 = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            n8 = L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
            y8b = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
            assert (n8 > 0) == expect_exact, (name, n8)
            np.testing.assert_array_equal(y8, y8b, err_msg=name + ": two runs of the 8-wave kernel differ")
            assert_close(y8, ref, c.tol, c.tol_max, what=name + " (8-wave kernel)")
            d = rel_l2(y8, y2)
            print(f"[parity] {name}: 8-wave vs 4-wave kernel rel_l2={d:.3e}, exact repeats {n8}")
            assert d <= 1e-3, (name, d)
    finally:
        L.tsd_debug_set_attn_qb(gpu_ctx.h, prev)


@pytest.mark.parametrize("scale", [0.02, 8.0])
def test_attention_8wave_kernel_reference_rounding_at_tiny_and_huge_scores(gpu_ctx, tsd_mod, scale):
    """The 8-wave kernel subtracts the softmax reference inside the QK^T MFMAs as (-ref / 32 in fp16) x 32 (kernels_attn8.hip): references
    below 1/16 snap to 0 (no fp16 subnormals in an MFMA operand), large ones are rounded to 11 bits.  q / k projections scaled by 0.02 (all
    scores within +-0.01: the snap) and by 8 (scores x 64: hundreds of log2 units, a one-hot softmax that must take the exact repeat) on an
    18-tile key loop: both kernels against the oracle."""
    from tsd._lib import lib
    L = lib()
    c = CASES["self_attention_d40_long_rising"]
    i = dict(c.build())
    wi = np.array(i["wi"], dtype=np.float32, copy=True)
    wi[: 2 * 320] *= np.float32(scale)
    i["wi"] = wi
    ref = np.asarray(c.oracle(i), dtype=np.float32)
    from util import rel_l2
    # scores x 64: the fp16 rounding of q and k (5e-4 each) is half a log2 unit of the exponent - near-ties of the one-hot softmax flip against
    # the fp32 oracle whatever the kernel; that leg is held to a loose bound against the oracle and to a tight one between the two kernels
    tol, tol_max = (c.tol, c.tol_max) if scale < 1.0 else (2e-2, 1.5e-1)
    prev = L.tsd_debug_set_attn_qb(gpu_ctx.h, 2)
    ys = {}
    try:
        for mode in (2, 3):
            L.tsd_debug_set_attn_qb(gpu_ctx.h, mode)
            L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
            y = np.asarray(c.device(tsd_mod, i), dtype=np.float32)
            n = L.tsd_debug_attn_exact_passes(gpu_ctx.h, 1)
            assert np.isfinite(y).all()
            assert (n > 0) == (scale > 1.0), (mode, scale, n)
            assert_close(y, ref, tol, tol_max, what=f"self-attention, q / k x {scale}, kernel mode {mode}")
            ys[mode] = y
    finally:
        L.tsd_debug_set_attn_qb(gpu_ctx.h, prev)
    d = rel_l2(ys[3], ys[2])
    print(f"[parity] q / k x {scale}: 8-wave vs 4-wave kernel rel_l2={d:.3e}")
    assert d <= 2e-3, d


@pytest.mark.parametrize("B,H,Cin,N,cfg,ref", [(2, 64, 64, 160, 30, 0), (2, 64, 640, 320, 30, 0), (1, 128, 128, 128, 32, 2),
                                               (1, 256, 128, 128, 32, 2), (2, 64, 128, 256, 32, 2)])
def test_halo_x_conv_order_matches_plain_tiles(gpu_ctx, B, H, Cin, N, cfg, ref):
    """The opt-in halo-x conv K order (kernels_gemm.hip HX, TSD_CONV_HALO: one staged tile per (kh, channel chunk) serves
    the three kw taps) against the plain tile configuration on the same synthetic problem: equal up to the fp32 summation
    order (bit-identical when there is a single channel chunk)."""
    import ctypes as C
    from tsd._lib import lib
    d, r = C.c_float(), C.c_float()
    assert lib().tsd_debug_gemm_check(gpu_ctx.h, 1, B, H, H, Cin, N, 1, 0, cfg, ref, C.byref(d), C.byref(r)) == 0
    assert r.value > 0.5 and d.value <= (0.0 if Cin == 64 else 2e-3 * r.value), (d.value, r.value)
    # an ineligible problem (stride 2) must be refused, not mis-addressed
    assert lib().tsd_debug_gemm_check(gpu_ctx.h, 1, B, H, H, Cin, N, 2, 0, cfg, ref, C.byref(d), C.byref(r)) != 0


@pytest.mark.parametrize("T", [1, 5, 63, 64, 65, 80])
def test_fused_attention_block_context_lengths(gpu_ctx, tsd_mod, T):
    """`Unet_Attention_Block` at C = 320 (the fused head / tail kernels) for context lengths around the tail's key-fragment
    boundaries: T < 64 masks whole key fragments, T = 64 leaves the fifth fragment empty, 65..80 fill it partly / fully
    (helpers/attention.mojo:105-115: softmax over the T context keys; the row sum comes from a ones row in the P.V MFMAs)."""
    from cases import _attn_params
    from oracle import models
    from util import TOL_BLOCK, TOL_BLOCK_MAX, randn
    nh, ne, H = 8, 40, 16
    C = nh * ne
    i = dict(x=randn(60, C, H, H), c=randn(61 + T, T, 768), P=_attn_params("a", C, 70))
    ref = np.asarray(models.unet_attention_block(i["P"], "a", i["x"], i["c"], nh, ne), np.float32)
    y = np.asarray(CASES["unet_attn_8x40"].device(tsd_mod, i), np.float32)
    assert_close(y, ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"unet_attn_8x40 with {T} context tokens")


@pytest.mark.parametrize("cin,cout,H", [(640, 1280, 16), (640, 640, 32)])
def test_optin_256_row_splitk_tiles_match_oracle(cin, cout, H, gpu_ctx, tsd_mod, monkeypatch):
    """`TSD_GEMM_SK256` (off by default, experiments/README.md): the split-K layers of the 16x16 level (bit 0) and the 640-wide
    convolutions of the 32x32 level (bit 1) on 256-row tiles with twice the K slices.  The option is read when a context is created;
    a context created with both bits on must still give the reference's residual block, and the default context must be untouched."""
    from cases import _res_params
    from oracle import models
    from tsd._lib import Context
    from util import TOL_BLOCK, TOL_BLOCK_MAX, randn
    x, time, P = randn(30, cin, H, H), randn(31, 1, 1280), _res_params("r", cin, cout, 40)

    def run(ctx):
        r = tsd_mod.Unet_Residual_Block(cin, cout, ctx=ctx)
        r.layer2.kernel, r.layer2.bias = P["r.layer2.kernel"], P["r.layer2.bias"]
        r.layer3.weight, r.layer3.bias = P["r.layer3.weight"], P["r.layer3.bias"]
        r.layer5.kernel, r.layer5.bias = P["r.layer5.kernel"], P["r.layer5.bias"]
        if cin != cout:
            r.layer6.kernel, r.layer6.bias = P["r.layer6.kernel"], P["r.layer6.bias"]
        return np.asarray(r.forward(x, time), np.float32)

    before = run(gpu_ctx)
    monkeypatch.setenv("TSD_GEMM_SK256", "3")
    ctx = Context(gpu_ctx.device)
    monkeypatch.delenv("TSD_GEMM_SK256")
    y = run(ctx)
    ref = np.asarray(models.unet_residual_block(P, "r", x, time, cin, cout), np.float32)
    assert_close(y, ref, TOL_BLOCK, TOL_BLOCK_MAX, what=f"res {cin}->{cout} @{H} with 256-row split-K tiles")
    assert_close(before, ref, TOL_BLOCK, TOL_BLOCK_MAX, what="default context")
    assert not np.array_equal(y, before)            # another summation tree really ran
    assert np.array_equal(run(gpu_ctx), before)     # ... and only in the context that asked for it
