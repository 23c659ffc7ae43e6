"""Parity cases shared by the golden-fixture generator, the CPU oracle test and the GPU tests.

Each case: build() -> inputs (seeded counter RNG, so nothing big is stored), oracle(inputs) -> reference,
device(tsd, inputs) -> result through the C ABI, and the stated tolerance (util.TOL_*).
Shapes cover the reference's own configurations (head dims 40/80/160, group sizes 10/20/30, stride-2,
the encoder's asymmetric pad, 77-token context) plus ragged edge cases (tails in M/N/K, odd sizes).
"""
import numpy as np

from oracle import models, ops
from util import TOL_BLOCK, TOL_BLOCK_MAX, TOL_OP, TOL_OP_MAX, randn, uni

CASES = {}


def case(name, tol=TOL_OP, tol_max=TOL_OP_MAX):
    def deco(cls):
        cls.name, cls.tol, cls.tol_max = name, tol, tol_max
        CASES[name] = cls
        return cls
    return deco


def _w(tag, *shape):
    fan_in = int(np.prod(shape[1:]))
    return uni(tag, 1.0 / np.sqrt(fan_in), *shape)


# ---- Conv2D -----------------------------------------------------------------------------------------
def _conv_case(name, C, O, k, H, W, pad, stride, I=None, bias=True):
    I = I or C

    @case(name)
    class _C:
        @staticmethod
        def build():
            return dict(x=randn(1, C, H, W), w=_w(2, O, I, k, k), b=randn(3, O) * 0.1 if bias else None)

        @staticmethod
        def oracle(i):
            return ops.conv2d(i["x"], i["w"], i["b"], padding=(pad, pad), stride=(stride, stride))

        @staticmethod
        def device(tsd, i):
            c = tsd.Conv2D(I, O, k, (pad, pad), (stride, stride))
            c.kernel = i["w"]
            c.bias = i["b"] if i["b"] is not None else np.zeros(O, np.float32)
            return c.forward(i["x"])
    return _C


_conv_case("conv3x3_64_64_s1", 64, 64, 3, 16, 16, 1, 1)
_conv_case("conv3x3_320_320_s2", 320, 320, 3, 16, 16, 1, 2)       # unet.layer4 shape class
_conv_case("conv3x3_ragged", 24, 20, 3, 13, 9, 1, 1)               # Cin/Cout/M tails
_conv_case("conv3x3_4_320", 4, 320, 3, 8, 8, 1, 1)                 # unet.layer1
_conv_case("conv3x3_320_4", 320, 4, 3, 8, 8, 1, 1)                 # final.layer2
_conv_case("conv3x3_128_3", 128, 3, 3, 16, 16, 1, 1)               # decoder l26
_conv_case("conv3x3_pad0", 64, 32, 3, 10, 10, 0, 1)
_conv_case("conv3x3_pad0_s2", 64, 32, 3, 11, 11, 0, 2)
_conv_case("conv1x1_640_320", 640, 320, 1, 8, 8, 0, 1)             # residual skip conv
_conv_case("conv1x1_4_4", 4, 4, 1, 8, 8, 0, 1)                     # decoder l1
_conv_case("conv3x3_first_channels_only", 96, 64, 3, 8, 8, 1, 1, I=64, bias=False)   # App.A D11


@case("pad_asymmetric")
class _Pad:
    @staticmethod
    def build():
        return dict(x=randn(4, 5, 7, 6))

    @staticmethod
    def oracle(i):
        return ops.pad(i["x"], (0, 1), (0, 1))  # vae.mojo:115-116

    @staticmethod
    def device(tsd, i):
        return tsd.pad(i["x"], (0, 1), (0, 1))
_Pad.tol, _Pad.tol_max = 0.0, 0.0


# ---- norms --------------------------------------------------------------------------------------------
def _gn_case(name, C, G, H, W, eps, Cn=None):
    Cn = Cn or C

    @case(name)
    class _G:
        @staticmethod
        def build():
            return dict(x=randn(5, C, H, W) * 2.0 + 0.5)

        @staticmethod
        def oracle(i):
            return ops.group_norm(i["x"], G, Cn, eps)

        @staticmethod
        def device(tsd, i):
            return tsd.GroupNorm(G, Cn, eps).forward(i["x"])
    return _G


_gn_case("groupnorm_320_32", 320, 32, 8, 8, 1e-5)      # 10 channels / group
_gn_case("groupnorm_960_32", 960, 32, 4, 4, 1e-5)      # 30 channels / group (concat width)
_gn_case("groupnorm_320_320", 320, 320, 8, 8, 1e-5)    # final layer: instance norm
_gn_case("groupnorm_128_16", 128, 16, 16, 16, 1e-5)    # VAE Res_Block
_gn_case("groupnorm_eps1e-6", 640, 32, 4, 4, 1e-6)     # attention block
_gn_case("groupnorm_partial_channels", 96, 4, 5, 7, 1e-5, Cn=64)
_gn_case("groupnorm_big_hw", 64, 32, 48, 40, 1e-5)     # several slabs


# extension (not reference behaviour): torch-style norms for real checkpoints, pinned against torch in test_oracle_pins.py
def _gn_torch_case(name, C, G, H, W, eps, affine=True, silu=False):
    @case(name)
    class _GT:
        @staticmethod
        def build():
            return dict(x=randn(31, C, H, W) * 2.0 + 0.5, w=1.0 + 0.3 * randn(32, C), b=0.2 * randn(33, C))

        @staticmethod
        def oracle(i):
            y = ops.group_norm_torch(i["x"], G, eps, i["w"] if affine else None, i["b"] if affine else None)
            return ops.silu(y) if silu else y

        @staticmethod
        def device(tsd, i):
            return tsd.ext.TorchGroupNorm(G, C, eps, i["w"] if affine else None, i["b"] if affine else None,
                                          silu=silu).forward(i["x"])
    return _GT


_gn_torch_case("groupnorm_torch_320_32", 320, 32, 8, 8, 1e-5)
_gn_torch_case("groupnorm_torch_silu_1280", 1280, 32, 4, 4, 1e-5, silu=True)
_gn_torch_case("groupnorm_torch_no_affine_big_hw", 64, 32, 48, 40, 1e-6, affine=False)


def _ln_torch_case(name, M, C, affine=True):
    @case(name)
    class _LT:
        @staticmethod
        def build():
            return dict(x=randn(34, M, C) * 3.0 + 1.0, w=1.0 + 0.3 * randn(35, C), b=0.2 * randn(36, C))

        @staticmethod
        def oracle(i):
            return ops.layer_norm_torch(i["x"], 1e-5, i["w"] if affine else None, i["b"] if affine else None)

        @staticmethod
        def device(tsd, i):
            return tsd.ext.TorchLayerNorm(C, 1e-5, i["w"] if affine else None, i["b"] if affine else None).forward(i["x"])
    return _LT


_ln_torch_case("layernorm_torch_320", 37, 320)
_ln_torch_case("layernorm_torch_768", 11, 768)
_ln_torch_case("layernorm_torch_1280_no_affine", 9, 1280, affine=False)


@case("layernorm_320")
class _LN:
    @staticmethod
    def build():
        return dict(x=randn(6, 37, 320) * 3.0 + 1.0)

    @staticmethod
    def oracle(i):
        return ops.layer_norm(i["x"])

    @staticmethod
    def device(tsd, i):
        return tsd.LayerNorm(320).forward(i["x"])


@case("layernorm_1280")
class _LN2(_LN):
    @staticmethod
    def build():
        return dict(x=randn(7, 9, 1280))

    @staticmethod
    def device(tsd, i):
        return tsd.LayerNorm(1280).forward(i["x"])


# ---- elementwise ----------------------------------------------------------------------------------------
@case("silu", tol=1e-6, tol_max=1e-6)
class _Silu:
    @staticmethod
    def build():
        return dict(x=randn(8, 3, 17, 5) * 4)

    @staticmethod
    def oracle(i):
        return ops.silu(i["x"])

    @staticmethod
    def device(tsd, i):
        return tsd.SiLU().forward(i["x"])


@case("gelu_tanh", tol=1e-6, tol_max=1e-6)
class _Gelu(_Silu):
    @staticmethod
    def oracle(i):
        return ops.gelu_tanh(i["x"])

    @staticmethod
    def device(tsd, i):
        return tsd.Gelu().forward(i["x"])


@case("upsample", tol=0.0, tol_max=0.0)
class _Up:
    @staticmethod
    def build():
        return dict(x=randn(9, 6, 5, 7))

    @staticmethod
    def oracle(i):
        return ops.upsample_nearest2x(i["x"])

    @staticmethod
    def device(tsd, i):
        return tsd.Upsample(1280).forward(i["x"])  # scale_factor argument ignored (App.A D1)


@case("softmax", tol=1e-5, tol_max=1e-5)
class _Sm:
    @staticmethod
    def build():
        return dict(x=randn(10, 8, 11, 77) * 3)

    @staticmethod
    def oracle(i):
        return ops.softmax_lastdim(i["x"])

    @staticmethod
    def device(tsd, i):
        return tsd.Softmax(i["x"], dim=2)


@case("time_embedding", tol=1e-5, tol_max=1e-5)
class _Te:
    @staticmethod
    def build():
        return dict(t=np.float32(980.0))

    @staticmethod
    def oracle(i):
        return ops.time_embedding(float(i["t"])).reshape(1, 1, 320)

    @staticmethod
    def device(tsd, i):
        return tsd.get_time_embedding(float(i["t"]))


# ---- Linear / matmul ----------------------------------------------------------------------------------------
def _lin_case(name, M, K, N, bias=True):
    @case(name)
    class _L:
        @staticmethod
        def build():
            return dict(x=randn(11, M, K), w=_w(12, N, K), b=randn(13, N) * 0.1 if bias else None)

        @staticmethod
        def oracle(i):
            return ops.linear(i["x"], i["w"], i["b"])

        @staticmethod
        def device(tsd, i):
            l = tsd.Linear(K, N, use_bias=bias)
            l.weight = i["w"]
            if bias:
                l.bias = i["b"]
            return l.forward(i["x"])
    return _L


_lin_case("linear_320_960", 96, 320, 960, bias=False)   # in_proj
_lin_case("linear_768_640", 77, 768, 640, bias=False)   # k_proj on the 77-token context
_lin_case("linear_ragged", 5, 100, 37)                  # K not a multiple of 64, N not a multiple of 4, M=5
_lin_case("linear_time", 1, 1280, 320)                  # time projection, M=1


@case("matmul_broadcast")
class _Mm:
    @staticmethod
    def build():
        return dict(a=randn(14, 3, 20, 48), b=randn(15, 1, 48, 24))

    @staticmethod
    def oracle(i):
        return ops.matmul(i["a"], i["b"])

    @staticmethod
    def device(tsd, i):
        return tsd.matmul(i["a"], i["b"])


@case("matmul_batched")
class _Mm2(_Mm):
    @staticmethod
    def build():
        return dict(a=randn(16, 2, 33, 70), b=randn(17, 2, 70, 18))


# ---- attention ------------------------------------------------------------------------------------------------
def _sa_case(name, T, D, H, in_bias, ramp=None, tol=TOL_OP, tol_max=TOL_OP_MAX, spike=None):
    @case(name, tol=tol, tol_max=tol_max)
    class _S:
        @staticmethod
        def build():
            x = randn(18, T, D)
            if ramp is not None:  # token magnitudes grow (or shrink) along the sequence: the score maxima move from key tile to key tile
                x = x * np.linspace(ramp[0], ramp[1], T, dtype=np.float32)[:, None]
            if spike is not None:  # ONE token far larger than the rest: its key overflows the optimistic pass of about half the query rows
                x[spike[0]] *= np.float32(spike[1])
            return dict(x=x, wi=_w(19, 3 * D, D), bi=randn(20, 3 * D) * 0.1 if in_bias else None,
                        wo=_w(21, D, D), bo=randn(22, D) * 0.1)

        @staticmethod
        def oracle(i):
            return ops.self_attention(i["x"], H, i["wi"], i["bi"], i["wo"], i["bo"])

        @staticmethod
        def device(tsd, i):
            a = tsd.Self_Attention(H, D, in_bias=in_bias)
            a.in_proj.weight, a.out_proj.weight, a.out_proj.bias = i["wi"], i["wo"], i["bo"]
            if in_bias:
                a.in_proj.bias = i["bi"]
            return a.forward(i["x"])
    return _S


def _sa_self_peaked_case(name, T, D, H, gain):
    """Self-attention whose key projection equals its query projection: every token's largest score is with ITSELF (|q_i|^2 /
    sqrt(d), ~25 nats above an average score at this gain) - the peaked-on-the-own-neighbourhood pattern of trained SD
    self-attention.  For queries beyond key tile 0 that score is > 20 log2 units above tile 0's row maximum."""
    @case(name, tol=TOL_OP, tol_max=TOL_OP_MAX)
    class _S:
        @staticmethod
        def build():
            wq = _w(31, D, D) * gain
            wi = np.concatenate([wq, wq, _w(32, D, D)], axis=0)
            return dict(x=randn(33, T, D), wi=wi, wo=_w(34, D, D), bo=randn(35, D) * 0.1)

        @staticmethod
        def oracle(i):
            return ops.self_attention(i["x"], H, i["wi"], None, i["wo"], i["bo"])

        @staticmethod
        def device(tsd, i):
            a = tsd.Self_Attention(H, D, in_bias=False)
            a.in_proj.weight, a.out_proj.weight, a.out_proj.bias = i["wi"], i["wo"], i["bo"]
            return a.forward(i["x"])
    return _S


_sa_self_peaked_case("self_attention_d40_self_peaked", 320, 320, 8, 3.2)
_sa_case("self_attention_d40", 256, 320, 8, False)    # level-0 shape class (S=256 here)
_sa_case("self_attention_d80", 64, 640, 8, False)
_sa_case("self_attention_d160", 64, 1280, 8, False)
_sa_case("self_attention_d40_ragged", 72, 320, 8, False)   # Tq/Tk tails inside one 64-key tile + second tile
_sa_case("self_attention_vae_1head", 64, 128, 1, True)     # VAE style: one head, biases on (unfused path)
_sa_case("self_attention_d80_T96", 96, 640, 8, False)      # fused q/k/v projection with a partial row tile (96 = 3 passes of 32 tokens)
_sa_case("self_attention_d80_T96_bias", 96, 640, 8, True)  # ... and the bias over all 3C columns, V rows included
# The fused core keeps a lazily updated softmax reference (kernels_attn.hip, TSD_ATTN_LAZY): these rows make the running maximum
# climb by 30-45 log2 units across three to six key tiles (and start far below / above zero), so the reference-move path runs.
_sa_case("self_attention_d40_rising_scores", 320, 320, 8, False, ramp=(0.25, 4.5), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d40_falling_scores", 328, 320, 8, False, ramp=(4.5, 0.25), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d80_rising_scores", 200, 640, 8, False, ramp=(0.25, 5.0), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d160_rising_scores", 136, 1280, 8, False, ramp=(0.25, 6.5), tol=5e-3, tol_max=1e-2)
# Round 5: the optimistic pass looks at the row sums every eighth key tile and leaves early.  Key loops long enough for that (17 - 18
# tiles): scores that rise all the way (the first look at tile 7 already finds the overflow), one huge key in tile 14 (found by the
# look at tile 15), one in the last, partial tile (only the final check can find it), and the d = 80 kernel.
_sa_case("self_attention_d40_long_rising", 1152, 320, 8, False, ramp=(0.25, 4.5), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d40_spike_tile14", 1096, 320, 8, False, spike=(14 * 64 + 9, 60.0), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d40_spike_last_tile", 1096, 320, 8, False, spike=(1091, 60.0), tol=5e-3, tol_max=1e-2)
_sa_case("self_attention_d80_long_rising", 1096, 640, 8, False, ramp=(0.25, 5.0), tol=5e-3, tol_max=1e-2)


def _ca_case(name, Tq, D, H, Tk=77, Dc=768):
    @case(name, tol=TOL_OP, tol_max=TOL_OP_MAX)
    class _Cc:
        @staticmethod
        def build():
            return dict(x=randn(23, Tq, D), c=randn(24, Tk, Dc), wq=_w(25, D, D), wk=_w(26, D, Dc), wv=_w(27, D, Dc),
                        wo=_w(28, D, D), bo=randn(29, D) * 0.1)

        @staticmethod
        def oracle(i):
            return ops.cross_attention(i["x"], i["c"], H, i["wq"], None, i["wk"], None, i["wv"], None, i["wo"], i["bo"])

        @staticmethod
        def device(tsd, i):
            a = tsd.Cross_Attention(H, D, Dc, in_bias=False)
            a.q_proj.weight, a.k_proj.weight, a.v_proj.weight = i["wq"], i["wk"], i["wv"]
            a.out_proj.weight, a.out_proj.bias = i["wo"], i["bo"]
            return a.forward(i["x"], i["c"])
    return _Cc


_ca_case("cross_attention_d40_T77", 64, 320, 8)
_ca_case("cross_attention_d160_T77", 16, 1280, 8)
_ca_case("cross_attention_d80_T5", 24, 640, 8, Tk=5, Dc=100)


# ---- blocks ------------------------------------------------------------------------------------------------------
def _res_params(prefix, cin, cout, tag):
    P = {prefix + ".layer2.kernel": _w(tag, cout, cin, 3, 3), prefix + ".layer2.bias": randn(tag + 1, cout) * 0.1,
         prefix + ".layer3.weight": _w(tag + 2, cout, 1280), prefix + ".layer3.bias": randn(tag + 3, cout) * 0.1,
         prefix + ".layer5.kernel": _w(tag + 4, cout, cout, 3, 3), prefix + ".layer5.bias": randn(tag + 5, cout) * 0.1,
         prefix + ".layer6.kernel": _w(tag + 6, cout, cin, 1, 1), prefix + ".layer6.bias": randn(tag + 7, cout) * 0.1}
    return P


def _unet_res_case(name, cin, cout, H, Cx=None):
    Cx = Cx or cin

    @case(name, tol=TOL_BLOCK, tol_max=TOL_BLOCK_MAX)
    class _R:
        @staticmethod
        def build():
            return dict(x=randn(30, Cx, H, H), time=randn(31, 1, 1280), P=_res_params("r", cin, cout, 40))

        @staticmethod
        def oracle(i):
            return models.unet_residual_block(i["P"], "r", i["x"], i["time"], cin, cout)

        @staticmethod
        def device(tsd, i):
            r = tsd.Unet_Residual_Block(cin, cout)
            P = i["P"]
            r.layer2.kernel, r.layer2.bias = P["r.layer2.kernel"], P["r.layer2.bias"]
            r.layer3.weight, r.layer3.bias = P["r.layer3.weight"], P["r.layer3.bias"]
            r.layer5.kernel, r.layer5.bias = P["r.layer5.kernel"], P["r.layer5.bias"]
            r.layer6.kernel, r.layer6.bias = P["r.layer6.kernel"], P["r.layer6.bias"]
            return r.forward(i["x"], i["time"])
    return _R


_unet_res_case("unet_res_320_320", 320, 320, 8)
_unet_res_case("unet_res_320_640", 320, 640, 8)
_unet_res_case("unet_res_dead_concat", 640, 320, 8, Cx=960)     # layer20: declared 640 of 960 channels (App.A D11)
_unet_res_case("unet_res_640_1280_16", 640, 1280, 16)           # 16x16 level: 2-way split-K conv2 with the fused 1x1 skip segment
_unet_res_case("unet_res_ragged_128_192", 128, 192, 12)         # M = 144, N = 192: partial tiles in both directions through the fused skip
_unet_res_case("unet_res_2560_1280_8", 2560, 1280, 8)           # 8x8 level: 8 K slices, the last one starts inside the skip segment


def _attn_params(prefix, C, tag, dctx=768):
    n = prefix
    return {n + ".layer2.kernel": _w(tag, C, C, 1, 1), n + ".layer2.bias": randn(tag + 1, C) * 0.1,
            n + ".layer4.in_proj.weight": _w(tag + 2, 3 * C, C),
            n + ".layer4.out_proj.weight": _w(tag + 3, C, C), n + ".layer4.out_proj.bias": randn(tag + 4, C) * 0.1,
            n + ".layer6.q_proj.weight": _w(tag + 5, C, C), n + ".layer6.k_proj.weight": _w(tag + 6, C, dctx),
            n + ".layer6.v_proj.weight": _w(tag + 7, C, dctx),
            n + ".layer6.out_proj.weight": _w(tag + 8, C, C), n + ".layer6.out_proj.bias": randn(tag + 9, C) * 0.1,
            n + ".layer8.weight": _w(tag + 10, 8 * C, C), n + ".layer8.bias": randn(tag + 11, 8 * C) * 0.1,
            n + ".layer9.weight": _w(tag + 12, C, 4 * C), n + ".layer9.bias": randn(tag + 13, C) * 0.1,
            n + ".layer10.kernel": _w(tag + 14, C, C, 1, 1), n + ".layer10.bias": randn(tag + 15, C) * 0.1}


def _unet_attn_case(name, nh, ne, H):
    C = nh * ne

    @case(name, tol=TOL_BLOCK, tol_max=TOL_BLOCK_MAX)
    class _A:
        @staticmethod
        def build():
            return dict(x=randn(60, C, H, H), c=randn(61, 77, 768), P=_attn_params("a", C, 70))

        @staticmethod
        def oracle(i):
            return models.unet_attention_block(i["P"], "a", i["x"], i["c"], nh, ne)

        @staticmethod
        def device(tsd, i):
            a = tsd.Unet_Attention_Block(nh, ne)
            P = i["P"]
            a.layer2.kernel, a.layer2.bias = P["a.layer2.kernel"], P["a.layer2.bias"]
            a.layer4.in_proj.weight = P["a.layer4.in_proj.weight"]
            a.layer4.out_proj.weight, a.layer4.out_proj.bias = P["a.layer4.out_proj.weight"], P["a.layer4.out_proj.bias"]
            a.layer6.q_proj.weight, a.layer6.k_proj.weight = P["a.layer6.q_proj.weight"], P["a.layer6.k_proj.weight"]
            a.layer6.v_proj.weight = P["a.layer6.v_proj.weight"]
            a.layer6.out_proj.weight, a.layer6.out_proj.bias = P["a.layer6.out_proj.weight"], P["a.layer6.out_proj.bias"]
            a.layer8.weight, a.layer8.bias = P["a.layer8.weight"], P["a.layer8.bias"]
            a.layer9.weight, a.layer9.bias = P["a.layer9.weight"], P["a.layer9.bias"]
            a.layer10.kernel, a.layer10.bias = P["a.layer10.kernel"], P["a.layer10.bias"]
            return a.forward(i["x"], i["c"])
    return _A


_unet_attn_case("unet_attn_8x40", 8, 40, 16)
_unet_attn_case("unet_attn_8x80", 8, 80, 8)
_unet_attn_case("unet_attn_8x160", 8, 160, 4)


def _vae_res_case(name, cin, cout, H):
    @case(name, tol=TOL_BLOCK, tol_max=TOL_BLOCK_MAX)
    class _V:
        @staticmethod
        def build():
            return dict(x=randn(90, cin, H, H), P={
                "v.conv1.kernel": _w(91, cout, cin, 3, 3), "v.conv1.bias": randn(92, cout) * 0.1,
                "v.conv2.kernel": _w(93, cout, cout, 3, 3), "v.conv2.bias": randn(94, cout) * 0.1,
                "v.res_conv_layer.kernel": _w(95, cout, cin, 1, 1), "v.res_conv_layer.bias": randn(96, cout) * 0.1})

        @staticmethod
        def oracle(i):
            return models.vae_res_block(i["P"], "v", i["x"], cin, cout)

        @staticmethod
        def device(tsd, i):
            r = tsd.Res_Block(cin, cout)
            P = i["P"]
            r.conv1.kernel, r.conv1.bias = P["v.conv1.kernel"], P["v.conv1.bias"]
            r.conv2.kernel, r.conv2.bias = P["v.conv2.kernel"], P["v.conv2.bias"]
            r.res_conv_layer.kernel, r.res_conv_layer.bias = P["v.res_conv_layer.kernel"], P["v.res_conv_layer.bias"]
            return r.forward(i["x"])
    return _V


_vae_res_case("vae_res_128_128", 128, 128, 16)
_vae_res_case("vae_res_256_128", 256, 128, 8)


@case("vae_attention_512", tol=TOL_BLOCK, tol_max=TOL_BLOCK_MAX)
class _VA:
    @staticmethod
    def build():
        C = 512
        return dict(x=randn(100, C, 8, 8), wi=_w(101, 3 * C, C), bi=randn(102, 3 * C) * 0.1, wo=_w(103, C, C),
                    bo=randn(104, C) * 0.1)

    @staticmethod
    def oracle(i):
        P = {"v.attention.in_proj.weight": i["wi"], "v.attention.in_proj.bias": i["bi"],
             "v.attention.out_proj.weight": i["wo"], "v.attention.out_proj.bias": i["bo"]}
        return models.vae_attention_block(P, "v", i["x"])

    @staticmethod
    def device(tsd, i):
        a = tsd.Attention_Block(512)
        a.attention.in_proj.weight, a.attention.in_proj.bias = i["wi"], i["bi"]
        a.attention.out_proj.weight, a.attention.out_proj.bias = i["wo"], i["bo"]
        return a.forward(i["x"])


@case("time_embedding_mlp", tol=TOL_OP, tol_max=TOL_OP_MAX)
class _TM:
    @staticmethod
    def build():
        return dict(t=ops.time_embedding(500.0), w1=_w(110, 1280, 320), b1=randn(111, 1280) * 0.1,
                    w2=_w(112, 1280, 1280), b2=randn(113, 1280) * 0.1)

    @staticmethod
    def oracle(i):
        P = {"t.layer1.weight": i["w1"], "t.layer1.bias": i["b1"], "t.layer2.weight": i["w2"], "t.layer2.bias": i["b2"]}
        return models.time_embedding_mlp(P, i["t"], "t").reshape(1, 1, 1280)

    @staticmethod
    def device(tsd, i):
        t = tsd.Time_Embedding(320)
        t.layer1.weight, t.layer1.bias, t.layer2.weight, t.layer2.bias = i["w1"], i["b1"], i["w2"], i["b2"]
        return t.forward(i["t"])


@case("unet_output_layer", tol=TOL_BLOCK, tol_max=TOL_BLOCK_MAX)
class _OL:
    """`UNet_Output_Layer` diffusion.mojo:275-291 on its own: GroupNorm(320 groups over 320 channels = a
    per-channel instance norm) -> SiLU -> Conv2D(320, 4, 3, padding 1)."""
    @staticmethod
    def build():
        return dict(x=randn(120, 320, 16, 16) * 1.7 + 0.3, w=_w(121, 4, 320, 3, 3), b=randn(122, 4) * 0.1)

    @staticmethod
    def oracle(i):
        return models.unet_output_layer({"final.layer2.kernel": i["w"], "final.layer2.bias": i["b"]}, i["x"], "final")

    @staticmethod
    def device(tsd, i):
        o = tsd.UNet_Output_Layer(320, 4)
        o.layer2.kernel, o.layer2.bias = i["w"], i["b"]
        return o.forward(i["x"])


# cases whose oracle output is stored in tests/golden/golden.npz (kept < 1 MB)
GOLDEN = ["conv3x3_64_64_s1", "conv3x3_320_320_s2", "conv3x3_ragged", "conv3x3_4_320", "conv3x3_320_4", "conv1x1_4_4",
          "conv3x3_pad0_s2", "pad_asymmetric", "groupnorm_320_32", "groupnorm_320_320", "groupnorm_partial_channels",
          "layernorm_320", "silu", "gelu_tanh", "upsample", "softmax", "time_embedding", "linear_ragged", "linear_time",
          "matmul_broadcast", "self_attention_d40_ragged", "self_attention_d160", "cross_attention_d40_T77",
          "cross_attention_d80_T5", "unet_res_320_640", "unet_res_dead_concat", "unet_attn_8x80", "vae_res_256_128",
          "vae_attention_512", "time_embedding_mlp"]
