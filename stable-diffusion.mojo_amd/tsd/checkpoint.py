"""Real-checkpoint import for the full-size UNet (extension; no reference counterpart - the
reference can only random-initialise).

A diffusers-layout SD-1.x UNet (`unet/diffusion_pytorch_model.safetensors`: UNet2DConditionModel with block_out_channels
(320, 640, 1280, 1280), 2 resnets per down block, 8 heads, cross_attention_dim 768) maps one-to-one onto model kind
"diffusion_sd15_torch": the 45 flat layers of `SD15_STEPS`, the reference's struct-field names, plus the per-channel norm
parameters.  Pure host code: safetensors is parsed here (8-byte little-endian header length, JSON header, raw
little-endian tensors; F32 / F16 / BF16), the tensors go through `tsd_model_set_param`.

Numerics: fp16 storage of weights and activations with fp32 accumulation; the torch UNet kind uses torch's exact GELU
in the GEGLU gate and per-channel affine norms with eps inside the root, like diffusers."""
import json
import struct

import numpy as np

# flat layer (1-based position in SD15_STEPS) -> diffusers module prefix
SD15_MODULES = (
    ["conv_in"]
    + [f"down_blocks.0.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "downsamplers.0.conv")]
    + [f"down_blocks.1.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "downsamplers.0.conv")]
    + [f"down_blocks.2.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "downsamplers.0.conv")]
    + ["down_blocks.3.resnets.0", "down_blocks.3.resnets.1"]
    + ["mid_block.resnets.0", "mid_block.attentions.0", "mid_block.resnets.1"]
    + [f"up_blocks.0.{m}" for m in ("resnets.0", "resnets.1", "resnets.2", "upsamplers.0.conv")]
    + [f"up_blocks.1.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "resnets.2", "attentions.2",
                                    "upsamplers.0.conv")]
    + [f"up_blocks.2.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "resnets.2", "attentions.2",
                                    "upsamplers.0.conv")]
    + [f"up_blocks.3.{m}" for m in ("resnets.0", "attentions.0", "resnets.1", "attentions.1", "resnets.2", "attentions.2")]
)
assert len(SD15_MODULES) == 45

# (our field suffix, diffusers suffix) inside one block; conv weights are `.kernel` on our side
_RES = [("layer1.weight", "norm1.weight"), ("layer1.bias", "norm1.bias"),
        ("layer2.kernel", "conv1.weight"), ("layer2.bias", "conv1.bias"),
        ("layer3.weight", "time_emb_proj.weight"), ("layer3.bias", "time_emb_proj.bias"),
        ("layer4.weight", "norm2.weight"), ("layer4.bias", "norm2.bias"),
        ("layer5.kernel", "conv2.weight"), ("layer5.bias", "conv2.bias")]
_RES_SKIP = [("layer6.kernel", "conv_shortcut.weight"), ("layer6.bias", "conv_shortcut.bias")]
_T = "transformer_blocks.0."
_ATTN = [("layer1.weight", "norm.weight"), ("layer1.bias", "norm.bias"),
         ("layer2.kernel", "proj_in.weight"), ("layer2.bias", "proj_in.bias"),
         ("layer3.weight", _T + "norm1.weight"), ("layer3.bias", _T + "norm1.bias"),
         ("layer4.out_proj.weight", _T + "attn1.to_out.0.weight"), ("layer4.out_proj.bias", _T + "attn1.to_out.0.bias"),
         ("layer5.weight", _T + "norm2.weight"), ("layer5.bias", _T + "norm2.bias"),
         ("layer6.q_proj.weight", _T + "attn2.to_q.weight"), ("layer6.k_proj.weight", _T + "attn2.to_k.weight"),
         ("layer6.v_proj.weight", _T + "attn2.to_v.weight"),
         ("layer6.out_proj.weight", _T + "attn2.to_out.0.weight"), ("layer6.out_proj.bias", _T + "attn2.to_out.0.bias"),
         ("layer7.weight", _T + "norm3.weight"), ("layer7.bias", _T + "norm3.bias"),
         ("layer8.weight", _T + "ff.net.0.proj.weight"), ("layer8.bias", _T + "ff.net.0.proj.bias"),
         ("layer9.weight", _T + "ff.net.2.weight"), ("layer9.bias", _T + "ff.net.2.bias"),
         ("layer10.kernel", "proj_out.weight"), ("layer10.bias", "proj_out.bias")]
_TOP = [("time_embed.layer1.weight", "time_embedding.linear_1.weight"), ("time_embed.layer1.bias", "time_embedding.linear_1.bias"),
        ("time_embed.layer2.weight", "time_embedding.linear_2.weight"), ("time_embed.layer2.bias", "time_embedding.linear_2.bias"),
        ("final.layer1.weight", "conv_norm_out.weight"), ("final.layer1.bias", "conv_norm_out.bias"),
        ("final.layer2.kernel", "conv_out.weight"), ("final.layer2.bias", "conv_out.bias")]

_DT = {"F32": (np.float32, 4), "F16": (np.float16, 2), "BF16": (np.uint16, 2), "F64": (np.float64, 8)}


def read_safetensors(path):
    """{name: float32 array} of every F32 / F16 / BF16 / F64 tensor in the file."""
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        header = json.loads(f.read(n).decode("utf-8"))
        data = f.read()
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        if meta["dtype"] not in _DT:
            raise ValueError(f"{name}: unsupported safetensors dtype {meta['dtype']}")
        dt, size = _DT[meta["dtype"]]
        lo, hi = meta["data_offsets"]
        count = int(np.prod(meta["shape"])) if meta["shape"] else 1
        if hi - lo != count * size:
            raise ValueError(f"{name}: {hi - lo} bytes for shape {meta['shape']} {meta['dtype']}")
        raw = np.frombuffer(data, dtype=dt, count=count, offset=lo)
        if meta["dtype"] == "BF16":
            raw = (raw.astype(np.uint32) << 16).view(np.float32)
        out[name] = np.ascontiguousarray(raw, dtype=np.float32).reshape(meta["shape"])
    return out


def write_safetensors(path, tensors, dtype="F32"):
    """Write {name: array} (tools and tests; F32, F16 or BF16 with round-to-nearest-even)."""
    header, blobs, off = {}, [], 0
    for name in sorted(tensors):
        a = np.ascontiguousarray(tensors[name], dtype=np.float32)
        if dtype == "F32":
            b = a.tobytes()
        elif dtype == "F16":
            b = a.astype(np.float16).tobytes()
        elif dtype == "BF16":
            u = a.view(np.uint32)
            b = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16).tobytes()
        else:
            raise ValueError(dtype)
        header[name] = {"dtype": dtype, "shape": list(a.shape), "data_offsets": [off, off + len(b)]}
        blobs.append(b)
        off += len(b)
    h = json.dumps(header, separators=(",", ":")).encode("utf-8")
    h += b" " * ((8 - len(h) % 8) % 8)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", len(h)))
        f.write(h)
        for b in blobs:
            f.write(b)


def _layer_kinds():
    """{flat layer position: "conv" | "res" | "attn"} from the library's own parameter inventory."""
    from .model import param_specs
    fields = {}
    for name, _, _, _ in param_specs("diffusion_sd15"):
        if name.startswith("unet.layer"):
            fields.setdefault(int(name.split(".")[1][5:]), set()).add(name.split(".", 2)[2])
    return {i: ("attn" if "layer4.in_proj.weight" in f else "res" if "layer3.weight" in f else "conv")
            for i, f in fields.items()}


def diffusers_sd15_unet_to_params(state, prefix=""):
    """diffusers UNet2DConditionModel state dict -> {our parameter name: array} for kind "diffusion_sd15_torch"."""
    g = lambda k: np.asarray(state[prefix + k], dtype=np.float32)  # noqa: E731
    out = {ours: g(theirs) for ours, theirs in _TOP}
    kinds = _layer_kinds()
    for i, mod in enumerate(SD15_MODULES, start=1):
        n, kind = f"unet.layer{i}", kinds[i]
        if kind == "conv":
            out[n + ".kernel"], out[n + ".bias"] = g(mod + ".weight"), g(mod + ".bias")
        elif kind == "res":
            for ours, theirs in _RES:
                out[f"{n}.{ours}"] = g(f"{mod}.{theirs}")
            if prefix + f"{mod}.conv_shortcut.weight" in state:
                for ours, theirs in _RES_SKIP:
                    out[f"{n}.{ours}"] = g(f"{mod}.{theirs}")
        else:
            for ours, theirs in _ATTN:
                a = g(f"{mod}.{theirs}")
                if ours.endswith(".kernel") and a.ndim == 2:  # SD-2.x style linear projection
                    a = a[:, :, None, None]
                out[f"{n}.{ours}"] = a
            t = f"{mod}.{_T}attn1."
            out[n + ".layer4.in_proj.weight"] = np.concatenate([g(t + "to_q.weight"), g(t + "to_k.weight"), g(t + "to_v.weight")])
    return out


def params_to_diffusers_sd15_unet(params, prefix=""):
    """Inverse of `diffusers_sd15_unet_to_params` (export; used by the round-trip test)."""
    out = {prefix + theirs: np.asarray(params[ours], np.float32) for ours, theirs in _TOP}
    kinds = _layer_kinds()
    for i, mod in enumerate(SD15_MODULES, start=1):
        n, kind = f"unet.layer{i}", kinds[i]
        if kind == "conv":
            out[prefix + mod + ".weight"], out[prefix + mod + ".bias"] = params[n + ".kernel"], params[n + ".bias"]
        elif kind == "res":
            for ours, theirs in _RES:
                out[prefix + f"{mod}.{theirs}"] = params[f"{n}.{ours}"]
            w = params.get(n + ".layer6.kernel")
            if w is not None and w.shape[0] != w.shape[1]:
                for ours, theirs in _RES_SKIP:
                    out[prefix + f"{mod}.{theirs}"] = params[f"{n}.{ours}"]
        else:
            for ours, theirs in _ATTN:
                out[prefix + f"{mod}.{theirs}"] = params[f"{n}.{ours}"]
            q, k, v = np.split(np.asarray(params[n + ".layer4.in_proj.weight"]), 3)
            t = prefix + f"{mod}.{_T}attn1."
            out[t + "to_q.weight"], out[t + "to_k.weight"], out[t + "to_v.weight"] = q, k, v
    return out


def load_sd15_unet(path, ctx=None, prefix=""):
    """tsd.Diffusion(variant="diffusion_sd15_torch") with the weights of a diffusers SD-1.x UNet safetensors file."""
    from .diffusion import Diffusion
    from .model import param_specs
    params = diffusers_sd15_unet_to_params(read_safetensors(path), prefix)
    for name, shape, used, _ in param_specs("diffusion_sd15_torch"):
        if name not in params:
            if used:
                raise KeyError(f"checkpoint has no tensor for {name}")
            params[name] = np.zeros(shape, np.float32)  # fields the reference allocates but never reads
        elif tuple(params[name].shape) != tuple(shape):
            raise ValueError(f"{name}: checkpoint shape {params[name].shape}, model expects {shape}")
    return Diffusion(ctx=ctx, params=params, variant="diffusion_sd15_torch")


def hf_clip_text_to_params(state):
    """Hugging Face CLIPTextModel state dict (with or without the `text_model.` prefix; tensors or arrays) ->
    {our parameter name: array} for kind "clip_torch" (q/k/v stacked into the reference's in_proj)."""
    pre = "text_model." if any(k.startswith("text_model.") for k in state) else ""

    def g(k):
        v = state[pre + k]
        return np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v, dtype=np.float32)

    out = {"embedding.token.weight": g("embeddings.token_embedding.weight"),
           "embedding.position": g("embeddings.position_embedding.weight").reshape(-1),
           "layernorm.weight": g("final_layer_norm.weight"), "layernorm.bias": g("final_layer_norm.bias")}
    for i in range(12):
        h, n = f"encoder.layers.{i}.", f"player{i + 1}"
        out[n + ".layer2.in_proj.weight"] = np.concatenate([g(h + f"self_attn.{x}_proj.weight") for x in "qkv"])
        out[n + ".layer2.in_proj.bias"] = np.concatenate([g(h + f"self_attn.{x}_proj.bias") for x in "qkv"])
        for ours, theirs in (("layer2.out_proj", "self_attn.out_proj"), ("layer4", "mlp.fc1"), ("layer5", "mlp.fc2"),
                             ("layer1", "layer_norm1"), ("layer3", "layer_norm2")):
            out[f"{n}.{ours}.weight"], out[f"{n}.{ours}.bias"] = g(h + theirs + ".weight"), g(h + theirs + ".bias")
    return out


def load_clip_text(path_or_state, ctx=None):
    """tsd.CLIP(variant="clip_torch") from a CLIPTextModel safetensors file (e.g. `text_encoder/model.safetensors` of an
    SD-1.x repository) or an in-memory state dict."""
    from .clip import CLIP
    state = read_safetensors(path_or_state) if isinstance(path_or_state, str) else path_or_state
    return CLIP(ctx=ctx, params=hf_clip_text_to_params(state), variant="clip_torch")


# ---- VAE (diffusers AutoencoderKL) ---------------------------------------------------------------------------------
# layer position in DECODER_LAYERS / ENCODER_LAYERS (1-based) -> diffusers module; None = no parameters
VAE_DECODER_MODULES = {1: "post_quant_conv", 2: "decoder.conv_in", 3: "decoder.mid_block.resnets.0",
                       4: "decoder.mid_block.attentions.0", 5: "decoder.mid_block.resnets.1",
                       6: "decoder.up_blocks.0.resnets.0", 7: "decoder.up_blocks.0.resnets.1", 8: "decoder.up_blocks.0.resnets.2",
                       10: "decoder.up_blocks.0.upsamplers.0.conv",
                       11: "decoder.up_blocks.1.resnets.0", 12: "decoder.up_blocks.1.resnets.1", 13: "decoder.up_blocks.1.resnets.2",
                       15: "decoder.up_blocks.1.upsamplers.0.conv",
                       16: "decoder.up_blocks.2.resnets.0", 17: "decoder.up_blocks.2.resnets.1", 18: "decoder.up_blocks.2.resnets.2",
                       20: "decoder.up_blocks.2.upsamplers.0.conv",
                       21: "decoder.up_blocks.3.resnets.0", 22: "decoder.up_blocks.3.resnets.1", 23: "decoder.up_blocks.3.resnets.2",
                       24: "decoder.conv_norm_out", 26: "decoder.conv_out"}
VAE_ENCODER_MODULES = {1: "encoder.conv_in", 2: "encoder.down_blocks.0.resnets.0", 3: "encoder.down_blocks.0.resnets.1",
                       4: "encoder.down_blocks.0.downsamplers.0.conv",
                       5: "encoder.down_blocks.1.resnets.0", 6: "encoder.down_blocks.1.resnets.1",
                       7: "encoder.down_blocks.1.downsamplers.0.conv",
                       8: "encoder.down_blocks.2.resnets.0", 9: "encoder.down_blocks.2.resnets.1",
                       10: "encoder.down_blocks.2.downsamplers.0.conv",
                       11: "encoder.down_blocks.3.resnets.0", 12: "encoder.down_blocks.3.resnets.1",
                       13: "encoder.mid_block.resnets.0", 14: "encoder.mid_block.attentions.0", 15: "encoder.mid_block.resnets.1",
                       16: "encoder.conv_norm_out", 18: "encoder.conv_out", 19: "quant_conv"}
_VRES = [("group_norm1.weight", "norm1.weight"), ("group_norm1.bias", "norm1.bias"), ("conv1.kernel", "conv1.weight"),
         ("conv1.bias", "conv1.bias"), ("group_norm2.weight", "norm2.weight"), ("group_norm2.bias", "norm2.bias"),
         ("conv2.kernel", "conv2.weight"), ("conv2.bias", "conv2.bias")]
_VRES_SKIP = [("res_conv_layer.kernel", "conv_shortcut.weight"), ("res_conv_layer.bias", "conv_shortcut.bias")]


def _vae_kinds(which):
    from .model import param_specs
    fields = {}
    for name, _, _, _ in param_specs(which + "_torch"):
        i, f = name.split(".", 1)
        fields.setdefault(int(i[1:]), set()).add(f)
    return {i: ("attn" if "attention.in_proj.weight" in f else "res" if "conv1.kernel" in f else
                "conv" if "kernel" in f else "gn") for i, f in fields.items()}


def diffusers_vae_to_params(state, which):
    """diffusers AutoencoderKL state dict -> {our parameter name: array} for kind "decoder_torch" / "encoder_torch".
    Attention projections may be named to_q/to_k/to_v/to_out.0 (current) or query/key/value/proj_attn (older files);
    1x1-conv shaped attention weights (C, C, 1, 1) are flattened."""
    mods = VAE_DECODER_MODULES if which == "decoder" else VAE_ENCODER_MODULES
    g = lambda k: np.asarray(state[k], dtype=np.float32)  # noqa: E731
    out = {}
    for i, kind in _vae_kinds(which).items():
        n, mod = f"l{i}", mods[i]
        if kind == "conv":
            out[n + ".kernel"], out[n + ".bias"] = g(mod + ".weight"), g(mod + ".bias")
        elif kind == "gn":
            out[n + ".weight"], out[n + ".bias"] = g(mod + ".weight"), g(mod + ".bias")
        elif kind == "res":
            for ours, theirs in _VRES:
                out[f"{n}.{ours}"] = g(f"{mod}.{theirs}")
            if f"{mod}.conv_shortcut.weight" in state:
                for ours, theirs in _VRES_SKIP:
                    out[f"{n}.{ours}"] = g(f"{mod}.{theirs}")
        else:
            names = ("to_q", "to_k", "to_v", "to_out.0") if f"{mod}.to_q.weight" in state else ("query", "key", "value", "proj_attn")
            flat = lambda k: g(k).reshape(g(k).shape[0], -1)  # noqa: E731
            out[n + ".attention.in_proj.weight"] = np.concatenate([flat(f"{mod}.{x}.weight") for x in names[:3]])
            out[n + ".attention.in_proj.bias"] = np.concatenate([g(f"{mod}.{x}.bias") for x in names[:3]])
            out[n + ".attention.out_proj.weight"], out[n + ".attention.out_proj.bias"] = flat(f"{mod}.{names[3]}.weight"), g(f"{mod}.{names[3]}.bias")
            out[n + ".group_norm.weight"], out[n + ".group_norm.bias"] = g(mod + ".group_norm.weight"), g(mod + ".group_norm.bias")
    return out


def params_to_diffusers_vae(params, which):
    """Inverse of `diffusers_vae_to_params` (current diffusers attention names)."""
    mods = VAE_DECODER_MODULES if which == "decoder" else VAE_ENCODER_MODULES
    out = {}
    for i, kind in _vae_kinds(which).items():
        n, mod = f"l{i}", mods[i]
        if kind == "conv":
            out[mod + ".weight"], out[mod + ".bias"] = params[n + ".kernel"], params[n + ".bias"]
        elif kind == "gn":
            out[mod + ".weight"], out[mod + ".bias"] = params[n + ".weight"], params[n + ".bias"]
        elif kind == "res":
            for ours, theirs in _VRES:
                out[f"{mod}.{theirs}"] = params[f"{n}.{ours}"]
            w = params.get(n + ".res_conv_layer.kernel")
            if w is not None and w.shape[0] != w.shape[1]:
                for ours, theirs in _VRES_SKIP:
                    out[f"{mod}.{theirs}"] = params[f"{n}.{ours}"]
        else:
            q, k, v = np.split(np.asarray(params[n + ".attention.in_proj.weight"]), 3)
            bq, bk, bv = np.split(np.asarray(params[n + ".attention.in_proj.bias"]), 3)
            for x, w, b in (("to_q", q, bq), ("to_k", k, bk), ("to_v", v, bv)):
                out[f"{mod}.{x}.weight"], out[f"{mod}.{x}.bias"] = w, b
            out[mod + ".to_out.0.weight"], out[mod + ".to_out.0.bias"] = params[n + ".attention.out_proj.weight"], params[n + ".attention.out_proj.bias"]
            out[mod + ".group_norm.weight"], out[mod + ".group_norm.bias"] = params[n + ".group_norm.weight"], params[n + ".group_norm.bias"]
    return out


def load_vae(path_or_state, which="decoder", ctx=None):
    """tsd.Decoder / tsd.Encoder (variant "*_torch") from a diffusers AutoencoderKL safetensors file
    (`vae/diffusion_pytorch_model.safetensors`) or an in-memory state dict."""
    from .model import param_specs
    from .vae import Decoder, Encoder
    state = read_safetensors(path_or_state) if isinstance(path_or_state, str) else path_or_state
    params = diffusers_vae_to_params(state, which)
    for name, shape, used, _ in param_specs(which + "_torch"):
        if name not in params:
            if used:
                raise KeyError(f"checkpoint has no tensor for {name}")
            params[name] = np.zeros(shape, np.float32)
    return (Decoder if which == "decoder" else Encoder)(ctx=ctx, params=params, variant=which + "_torch")
