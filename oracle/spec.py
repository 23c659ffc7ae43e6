"""Parameter inventory of the reference models, in struct-field DFS order.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Independent restatement of the layer lists in
`diffusion.mojo:175-201,299-302` and `vae.mojo:94-112,194-219`; the product has its own table
(csrc/model_spec.cpp, exported through `tsd_model_param_info`) and tests assert both agree.

Every learnable field the reference allocates is listed, including the ones its forward never
uses (`used=False`): `Linear.bias` when `use_bias=False` (helpers/utils.mojo:1939), the 1x1
skip conv of a residual block with cin == cout (diffusion.mojo:42, vae.mojo:46).

Synthetic init: conv kernel U(+-1/sqrt(cin*k*k)) (helpers/utils.mojo:1722-1724),
conv bias 0 (:1717; App.A D17), linear weight and bias U(+-1/sqrt(in)) (App.A D18).
"""
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from . import rng


@dataclass(frozen=True)
class Param:
    name: str
    shape: Tuple[int, ...]
    kind: str        # "conv_w" | "conv_b" | "lin_w" | "lin_b"
    bound: float     # uniform init bound (0 => zeros)
    used: bool = True
    offset: float = 0.0  # synthetic init = offset + U(+-bound)

    @property
    def numel(self):
        return int(np.prod(self.shape))


def _conv(out: List[Param], name, cin, cout, k, used=True):
    out.append(Param(name + ".kernel", (cout, cin, k, k), "conv_w", 1.0 / np.sqrt(cin * k * k), used))
    out.append(Param(name + ".bias", (cout,), "conv_b", 0.0, used))


def _lin(out: List[Param], name, fin, fout, use_bias=True, used=True):
    b = 1.0 / np.sqrt(fin)
    out.append(Param(name + ".weight", (fout, fin), "lin_w", b, used))
    out.append(Param(name + ".bias", (fout,), "lin_b", b, used and use_bias))


def _unet_res(out, name, cin, cout):  # diffusion.mojo:34-42
    _conv(out, name + ".layer2", cin, cout, 3)
    _lin(out, name + ".layer3", 1280, cout)
    _conv(out, name + ".layer5", cout, cout, 3)
    _conv(out, name + ".layer6", cin, cout, 1, used=(cin != cout))


def _unet_attn(out, name, n_head, n_embed, d_ctx=768):  # diffusion.mojo:87-98
    C = n_head * n_embed
    _conv(out, name + ".layer2", C, C, 1)
    _lin(out, name + ".layer4.in_proj", C, 3 * C, use_bias=False)
    _lin(out, name + ".layer4.out_proj", C, C)
    _lin(out, name + ".layer6.q_proj", C, C, use_bias=False)
    _lin(out, name + ".layer6.k_proj", d_ctx, C, use_bias=False)
    _lin(out, name + ".layer6.v_proj", d_ctx, C, use_bias=False)
    _lin(out, name + ".layer6.out_proj", C, C)
    _lin(out, name + ".layer8", C, 8 * C)
    _lin(out, name + ".layer9", 4 * C, C)
    _conv(out, name + ".layer10", C, C, 1)


# (kind, args) per UNet layer, diffusion.mojo:177-201
UNET_LAYERS = [
    ("conv", (4, 320, 3, 1)),      # layer1
    ("res", (320, 320)),           # layer2
    ("attn", (8, 40)),             # layer3
    ("conv", (320, 320, 3, 2)),    # layer4 (stride 2)
    ("res", (320, 640)),           # layer5
    ("attn", (8, 80)),             # layer6
    ("conv", (640, 640, 3, 2)),    # layer7 (stride 2)
    ("res", (640, 1280)),          # layer8
    ("attn", (8, 160)),            # layer9
    ("res", (2560, 1280)),         # layer10
    ("attn", (8, 160)),            # layer11
    ("res", (1920, 1280)),         # layer12
    ("attn", (8, 160)),            # layer13
    ("up", ()),                    # layer14
    ("res", (1280, 640)),          # layer15 (declared cin < concatenated channels; App.A D11)
    ("attn", (8, 80)),             # layer16
    ("res", (960, 640)),           # layer17
    ("attn", (8, 80)),             # layer18
    ("up", ()),                    # layer19
    ("res", (640, 320)),           # layer20 (App.A D11)
    ("attn", (8, 40)),             # layer21
    ("res", (640, 320)),           # layer22
    ("attn", (8, 40)),             # layer23
]


def diffusion_params() -> List[Param]:
    """`Diffusion` diffusion.mojo:299-302."""
    out: List[Param] = []
    _lin(out, "time_embed.layer1", 320, 1280)
    _lin(out, "time_embed.layer2", 1280, 1280)
    for i, (kind, a) in enumerate(UNET_LAYERS, start=1):
        name = f"unet.layer{i}"
        if kind == "conv":
            _conv(out, name, a[0], a[1], a[2])
        elif kind == "res":
            _unet_res(out, name, *a)
        elif kind == "attn":
            _unet_attn(out, name, *a)
    _conv(out, "final.layer2", 320, 4, 3)
    return out


# Full-size (SD-1.5-sized, 860 M parameter) UNet of BASELINE.json configs[4]: the 12-encoder / bottleneck / 12-decoder
# layout the reference's 23-layer list was trimmed from, built from the reference's own blocks.  NOT defined by the
# reference: parity for it is oracle-vs-product only.  (kind, args, flags): "push" = output
# kept as a skip, "pop" = input is concat(x, most recent skip); "upconv" = nearest 2x upsample then Conv3x3.
def _full_unet_steps():
    R = lambda ci, co, f="": ("res", (ci, co), f)
    A = lambda dh, f="": ("attn", (8, dh), f)
    return [
        ("conv", (4, 320, 3, 1), "push"),
        R(320, 320), A(40, "push"), R(320, 320), A(40, "push"), ("conv", (320, 320, 3, 2), "push"),
        R(320, 640), A(80, "push"), R(640, 640), A(80, "push"), ("conv", (640, 640, 3, 2), "push"),
        R(640, 1280), A(160, "push"), R(1280, 1280), A(160, "push"), ("conv", (1280, 1280, 3, 2), "push"),
        R(1280, 1280, "push"), R(1280, 1280, "push"),
        R(1280, 1280), A(160), R(1280, 1280),
        R(2560, 1280, "pop"), R(2560, 1280, "pop"), R(2560, 1280, "pop"), ("upconv", (1280, 1280, 3, 1), ""),
        R(2560, 1280, "pop"), A(160), R(2560, 1280, "pop"), A(160), R(1920, 1280, "pop"), A(160),
        ("upconv", (1280, 1280, 3, 1), ""),
        R(1920, 640, "pop"), A(80), R(1280, 640, "pop"), A(80), R(960, 640, "pop"), A(80),
        ("upconv", (640, 640, 3, 1), ""),
        R(960, 320, "pop"), A(40), R(640, 320, "pop"), A(40), R(640, 320, "pop"), A(40),
    ]


FULL_UNET_STEPS = _full_unet_steps()


def diffusion_sd15_params() -> List[Param]:
    out: List[Param] = []
    _lin(out, "time_embed.layer1", 320, 1280)
    _lin(out, "time_embed.layer2", 1280, 1280)
    for i, (kind, a, _) in enumerate(FULL_UNET_STEPS, start=1):
        name = f"unet.layer{i}"
        if kind in ("conv", "upconv"):
            _conv(out, name, a[0], a[1], a[2])
        elif kind == "res":
            _unet_res(out, name, *a)
        elif kind == "attn":
            _unet_attn(out, name, *a)
    _conv(out, "final.layer2", 320, 4, 3)
    return out


def _norm(out, name, C):  # torch-norm extension: per-channel weight (1 + U(+-0.3)) and bias (U(+-0.2))
    out.append(Param(name + ".weight", (C,), "lin_b", 0.3, True, 1.0))
    out.append(Param(name + ".bias", (C,), "lin_b", 0.2))


def diffusion_sd15_torch_params() -> List[Param]:
    """Full-size UNet with PyTorch norm semantics (extension): the kind-5 list followed by
    weight/bias of every GroupNorm / LayerNorm, named by the norm's field position in the reference struct."""
    out = diffusion_sd15_params()
    for i, (kind, a, _) in enumerate(FULL_UNET_STEPS, start=1):
        name = f"unet.layer{i}"
        if kind == "res":
            _norm(out, name + ".layer1", a[0])
            _norm(out, name + ".layer4", a[1])
        elif kind == "attn":
            C = a[0] * a[1]
            for n in (1, 3, 5, 7):
                _norm(out, f"{name}.layer{n}", C)
    _norm(out, "final.layer1", 320)
    return out


def _vae_res(out, name, cin, cout):  # vae.mojo:39-46
    _conv(out, name + ".conv1", cin, cout, 3)
    _conv(out, name + ".conv2", cout, cout, 3)
    _conv(out, name + ".res_conv_layer", cin, cout, 1, used=(cin != cout))


def _vae_attn(out, name, C):  # vae.mojo:9-11
    _lin(out, name + ".attention.in_proj", C, 3 * C)
    _lin(out, name + ".attention.out_proj", C, C)


# vae.mojo:194-219
DECODER_LAYERS = [
    ("conv", (4, 4, 1)), ("conv", (4, 512, 3)), ("res", (512, 512)), ("attn", (512,)),
    ("res", (512, 512)), ("res", (512, 512)), ("res", (512, 512)), ("res", (512, 512)),
    ("up", ()), ("conv", (512, 512, 3)), ("res", (512, 512)), ("res", (512, 512)), ("res", (512, 512)),
    ("up", ()), ("conv", (512, 512, 3)), ("res", (512, 256)), ("res", (256, 256)), ("res", (256, 256)),
    ("up", ()), ("conv", (256, 256, 3)), ("res", (256, 128)), ("res", (128, 128)), ("res", (128, 128)),
    ("gn", (32, 128)), ("silu", ()), ("conv", (128, 3, 3)),
]

# vae.mojo:94-112 ; "conv_s2" = stride-2 3x3 conv after the asymmetric (0,1),(0,1) pad (:115-116)
ENCODER_LAYERS = [
    ("conv", (3, 128, 3)), ("res", (128, 128)), ("res", (128, 128)), ("conv_s2", (128, 128, 3)),
    ("res", (128, 256)), ("res", (256, 256)), ("conv_s2", (256, 256, 3)),
    ("res", (256, 512)), ("res", (512, 512)), ("conv_s2", (512, 512, 3)),
    ("res", (512, 512)), ("res", (512, 512)), ("res", (512, 512)), ("attn", (512,)), ("res", (512, 512)),
    ("gn", (32, 512)), ("silu", ()), ("conv", (512, 8, 3)), ("conv", (8, 8, 1)),
]


def _vae_params(layers) -> List[Param]:
    out: List[Param] = []
    for i, (kind, a) in enumerate(layers, start=1):
        name = f"l{i}"
        if kind in ("conv", "conv_s2"):
            _conv(out, name, *a)
        elif kind == "res":
            _vae_res(out, name, *a)
        elif kind == "attn":
            _vae_attn(out, name, *a)
    return out


def decoder_params() -> List[Param]:
    return _vae_params(DECODER_LAYERS)


def encoder_params() -> List[Param]:
    return _vae_params(ENCODER_LAYERS)


def clip_params() -> List[Param]:
    """`CLIP.__init__` clip.mojo:74-88: ClipEmbedding(49408, 768, 77) + 12 x ClipPlayer(12, 768) + LayerNorm(768)
    (the reference's LayerNorm has no parameters, helpers/utils.mojo:2052-2061).

    Synthetic init (weights are inputs, App.A rule 3): the token table is N(0,1) in the reference
    (`init_weights_normal(0,1)` helpers/utils.mojo:2025) -> U(+-sqrt(3)) here (same variance, counter RNG); the
    position table is zero-initialised there (clip.mojo:13-15) -> U(+-0.02) here so parity tests exercise the add."""
    out: List[Param] = []
    out.append(Param("embedding.token.weight", (49408, 768), "lin_w", float(np.sqrt(3.0))))
    out.append(Param("embedding.position", (77 * 768,), "lin_b", 0.02))
    for i in range(1, 13):
        n = f"player{i}"
        _lin(out, n + ".layer2.in_proj", 768, 3 * 768)   # Self_Attention(12, 768): in_bias defaults to True
        _lin(out, n + ".layer2.out_proj", 768, 768)
        _lin(out, n + ".layer4", 768, 4 * 768)
        _lin(out, n + ".layer5", 4 * 768, 768)
    return out


def _vae_torch_params(layers) -> List[Param]:
    """VAE graph with the trained VAE's norms (extension): the reference list + weight/bias of every GroupNorm."""
    out = _vae_params(layers)
    for i, (kind, a) in enumerate(layers, start=1):
        name = f"l{i}"
        if kind == "res":
            _norm(out, name + ".group_norm1", a[0])
            _norm(out, name + ".group_norm2", a[1])
        elif kind == "attn":
            _norm(out, name + ".group_norm", a[0])
        elif kind == "gn":
            _norm(out, name, a[1])
    return out


def decoder_torch_params() -> List[Param]:
    return _vae_torch_params(DECODER_LAYERS)


def encoder_torch_params() -> List[Param]:
    return _vae_torch_params(ENCODER_LAYERS)


def clip_torch_params() -> List[Param]:
    """CLIP text encoder with torch LayerNorms (extension): the kind-4 list + weight/bias of every LayerNorm."""
    out = clip_params()
    for i in range(1, 13):
        _norm(out, f"player{i}.layer1", 768)
        _norm(out, f"player{i}.layer3", 768)
    _norm(out, "layernorm", 768)
    return out


MODEL_IDS = {"diffusion": 1, "decoder": 2, "encoder": 3, "clip": 4, "diffusion_sd15": 5, "diffusion_sd15_torch": 6, "clip_torch": 7,
             "decoder_torch": 8, "encoder_torch": 9}


def tensor_id(model: str, index: int) -> int:
    """RNG tensor id of parameter `index` of `model` (shared convention with the product)."""
    return MODEL_IDS[model] * 4096 + index


def init_params(model: str, seed: int, only_used=False):
    """Generate the synthetic weights of `model` ('diffusion'|'decoder'|'encoder'|'clip'|'diffusion_sd15') -> {name: array}."""
    plist = {"diffusion": diffusion_params, "decoder": decoder_params, "encoder": encoder_params, "clip": clip_params,
             "diffusion_sd15": diffusion_sd15_params, "diffusion_sd15_torch": diffusion_sd15_torch_params, "clip_torch": clip_torch_params,
             "decoder_torch": decoder_torch_params, "encoder_torch": encoder_torch_params}[model]()
    out = {}
    for i, p in enumerate(plist):
        if only_used and not p.used:
            continue
        if p.bound == 0.0:
            out[p.name] = np.zeros(p.shape, dtype=np.float32)
        else:
            w = rng.uniform(seed, tensor_id(model, i), p.numel, p.bound).reshape(p.shape)
            out[p.name] = (w + np.float32(p.offset)) if p.offset else w
    return out
