"""tsd - Python host side of libtsd.so, mirroring the reference's Mojo modules for the hot path.

  tsd.utils      <-> helpers/utils.mojo     Conv2D, GroupNorm, SiLU, Gelu, Linear, Upsample, LayerNorm, Softmax, ...
  tsd.attention  <-> helpers/attention.mojo Self_Attention, Cross_Attention
  tsd.diffusion  <-> diffusion.mojo         Time_Embedding, Unet_Residual_Block, Unet_Attention_Block, UNet, Diffusion
  tsd.vae        <-> vae.mojo               Attention_Block, Res_Block, Decoder, Encoder
  tsd.sampler    <-> sampler.mojo           DDPMSampler
  tsd.clip       <-> clip.mojo              CLIP text encoder (token ids -> context)
  tsd.tokenizer  <-> helpers/utils.mojo     Tokenizer, bpe_encode (host-only logic inside libtsd)
  tsd.pipeline   <-> pipeline.mojo          generate (hot loop; the context embedding is an input)
  tsd.ext        (no reference counterpart) torch-style GroupNorm / LayerNorm for real checkpoints

Every forward() is a call through the C ABI in include/tsd.h into hand-written HIP kernels for gfx950.
There is no CPU fallback.
"""
from . import _lib, rng  # noqa: F401
from . import ext  # noqa: F401  (non-reference extensions: torch-style norms)
from ._lib import Context, TsdError, default_context, set_default_context, set_strict  # noqa: F401
from .model import Model, Session, flop_count, param_specs  # noqa: F401
from .utils import (Conv2D, Gelu, GroupNorm, LayerNorm, Linear, SiLU, Softmax, Upsample, concat,  # noqa: F401
                    get_time_embedding, matmul, pad, rescale)
from .attention import Cross_Attention, Self_Attention  # noqa: F401
from .diffusion import (Diffusion, Time_Embedding, UNet, UNet_Output_Layer, Unet_Attention_Block,  # noqa: F401
                        Unet_Residual_Block)
from .vae import Attention_Block, Decoder, Encoder, Res_Block  # noqa: F401
from .clip import CLIP  # noqa: F401
from .tokenizer import Tokenizer, process_prompt  # noqa: F401
from .sampler import DDPMSampler  # noqa: F401
from .pipeline import encode_prompts, generate  # noqa: F401
