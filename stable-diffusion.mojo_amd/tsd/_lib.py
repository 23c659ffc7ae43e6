"""ctypes binding of libtsd.so (include/tsd.h) - the only way the Python host reaches the GPU.

There is no CPU fallback: if the shared library is missing or no gfx950 device is visible every
compute call fails loudly.  Shape errors follow the reference's convention (print a message and
return a null matrix, e.g. helpers/utils.mojo:1955-1957) unless `set_strict(True)`.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TSD_LIB names another build of the same library (the -DTSD_JITTER hazard-hunting build of `make jitter`); never a different backend
LIB_PATH = os.environ.get("TSD_LIB") or os.path.join(os.path.dirname(_HERE), "lib", "libtsd.so")

TSD_OK, TSD_E_ARG, TSD_E_SHAPE, TSD_E_ALLOC, TSD_E_HIP, TSD_E_RCCL, TSD_E_STATE, TSD_E_NONFINITE = 0, -1, -2, -3, -4, -5, -6, -7
# Kernel arguments in device memory instead of host-coherent memory: every kernel starts by reading its ~200 B argument
# block, and fetching it across the host link costs ~2 us per launch (measured: 169 -> 177 steps/s).  The HIP runtime
# reads the switch when it initialises, so it goes into the environment as soon as this package is imported - before
# libtsd.so (or torch, in the multi-GPU bench) makes the process's first HIP call.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

MODEL_DIFFUSION, MODEL_DECODER, MODEL_ENCODER, MODEL_CLIP, MODEL_DIFFUSION_SD15, MODEL_DIFFUSION_SD15_TORCH = 1, 2, 3, 4, 5, 6
MODEL_CLIP_TORCH, MODEL_DECODER_TORCH, MODEL_ENCODER_TORCH = 7, 8, 9


class TsdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libtsd error {code}: {msg}")
        self.code = code


_lib = None
_strict = False
fp = C.POINTER(C.c_float)
vp = C.c_void_p


def set_strict(flag: bool):
    """strict=True raises TsdError on every non-zero status; default mirrors the reference
    (print + null matrix) for TSD_E_SHAPE only."""
    global _strict
    _strict = bool(flag)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsdError(TSD_E_HIP, f"{LIB_PATH} not built - run __graft_entry__.build() (hipcc, gfx950); "
                                      "there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(l):
    i, i64, f, u64, sz = C.c_int, C.c_int64, C.c_float, C.c_uint64, C.c_size_t
    pp = C.POINTER(vp)
    sig = {
        "tsd_version": ([], i), "tsd_last_error": ([], C.c_char_p), "tsd_device_count": ([], i),
        "tsd_ctx_create": ([i, pp], i), "tsd_ctx_destroy": ([vp], i), "tsd_ctx_synchronize": ([vp], i),
        "tsd_ctx_timer_start": ([vp], i), "tsd_ctx_timer_stop": ([vp, fp], i),
        "tsd_ctx_profile_begin": ([vp], i), "tsd_ctx_profile_records": ([vp, C.POINTER(i), fp, i], i), "tsd_ctx_profile_end": ([vp, fp, C.POINTER(i), i], i),
        "tsd_conv2d_f32": ([vp, fp, i, i, i, fp, fp, i, i, i, i, i, i, i, fp], i),
        "tsd_pad_f32": ([vp, fp, i, i, i, i, i, i, i, fp], i),
        "tsd_groupnorm_f32": ([vp, fp, i, i, i, i, i, f, f, fp], i),
        "tsd_layernorm_f32": ([vp, fp, i, i, f, fp], i),
        "tsd_groupnorm_affine_f32": ([vp, fp, i, i, i, i, f, fp, fp, i, fp], i),
        "tsd_layernorm_affine_f32": ([vp, fp, i, i, f, fp, fp, fp], i),
        "tsd_silu_f32": ([vp, fp, i64, fp], i), "tsd_gelu_tanh_f32": ([vp, fp, i64, fp], i),
        "tsd_rescale_images_f32": ([vp, fp, i64, fp], i),
        "tsd_linear_f32": ([vp, fp, i, i, fp, fp, i, fp], i),
        "tsd_matmul_f32": ([vp, fp, fp, i, i, i, i, i, fp], i),
        "tsd_upsample_nearest2x_f32": ([vp, fp, i, i, i, fp], i),
        "tsd_softmax_lastdim_f32": ([vp, fp, i64, i, fp], i),
        "tsd_self_attention_f32": ([vp, fp, i, i, i, fp, fp, fp, fp, i, fp], i),
        "tsd_cross_attention_f32": ([vp, fp, i, i, fp, i, i, i, fp, fp, fp, fp, fp, fp, fp, fp, fp], i),
        "tsd_time_embedding_f32": ([vp, f, fp], i),
        "tsd_time_embedding_mlp_f32": ([vp, fp, fp, fp, fp, fp, fp], i),
        "tsd_unet_residual_block_f32": ([vp, fp, i, i, i, fp, i, i, fp, fp, fp, fp, fp, fp, fp, fp, fp], i),
        "tsd_unet_attention_block_f32": ([vp, fp, i, i, i, i, fp, i, i, C.POINTER(fp), i, fp], i),
        "tsd_vae_res_block_f32": ([vp, fp, i, i, i, i, fp, fp, fp, fp, fp, fp, fp], i),
        "tsd_vae_attention_block_f32": ([vp, fp, i, i, i, fp, fp, fp, fp, fp], i),
        "tsd_model_param_count": ([i], i),
        "tsd_model_param_info": ([i, i, C.c_char_p, i, C.POINTER(i64), C.POINTER(i), C.POINTER(i), fp], i),
        "tsd_model_create": ([vp, i, pp], i), "tsd_model_destroy": ([vp], i),
        "tsd_model_set_param": ([vp, i, fp, i64], i), "tsd_model_init_random": ([vp, u64], i),
        "tsd_model_packed_blob": ([vp, pp, C.POINTER(sz)], i), "tsd_model_mark_loaded": ([vp], i), "tsd_model_prepare": ([vp], i),
        "tsd_diffusion_forward": ([vp, fp, fp, fp, i, i, i, fp], i),
        "tsd_decoder_forward": ([vp, fp, i, i, fp], i),
        "tsd_encoder_forward": ([vp, fp, fp, i, i, fp], i),
        "tsd_clip_forward": ([vp, C.POINTER(C.c_int32), i, i, fp], i),
        "tsd_tokenizer_create": ([C.c_char_p, i, C.POINTER(vp)], i),
        "tsd_tokenizer_create_from_memory": ([C.c_char_p, C.c_size_t, i, C.POINTER(vp)], i),
        "tsd_tokenizer_destroy": ([vp], i),
        "tsd_tokenizer_find": ([vp, C.c_char_p], i),
        "tsd_tokenizer_token": ([vp, i, C.c_char_p, i, fp], i),
        "tsd_tokenizer_encode": ([vp, C.c_char_p, C.POINTER(C.c_int32), i, C.POINTER(C.c_int), C.POINTER(C.c_int)], i),
        "tsd_session_create": ([vp, vp, i, i, i, i, pp], i), "tsd_session_destroy": ([vp], i),
        "tsd_session_set_schedule": ([vp, i, i, i], i), "tsd_session_num_steps": ([vp], i),
        "tsd_session_timestep": ([vp, i], i),
        "tsd_session_upload": ([vp, fp, fp, fp, fp, f], i), "tsd_session_step": ([vp, i], i),
        "tsd_session_add_noise": ([vp, i, fp], i), "tsd_session_decode": ([vp], i),
        "tsd_session_download_latents": ([vp, fp], i), "tsd_session_download_images": ([vp, i, fp], i),
        "tsd_dist_unique_id": ([vp], i), "tsd_dist_init": ([vp, i, i, vp], i),
        "tsd_dist_broadcast_weights": ([vp, i], i), "tsd_dist_finalize": ([vp], i),
        "tsd_dist_comm_count": ([vp, C.POINTER(i)], i),
        "tsd_flop_count": ([i, i, i], C.c_double),
        "tsd_debug_splitk_errors": ([vp], i),
        "tsd_debug_xcd_round_robin": ([], i),
        "tsd_debug_nonfinite_count": ([vp, i], i),
        "tsd_debug_set_fused_attention": ([vp, i], i),
        "tsd_debug_gemm_bench": ([vp, i, i, i, i, i, i, i, i, i, i, fp], i),
        "tsd_debug_attn_bench": ([vp, i, i, i, i, i, i, fp], i),
        "tsd_debug_attn_exact_passes": ([vp, i], i),
        "tsd_debug_set_attn_qb": ([vp, i], i),
        "tsd_debug_set_attn_diag": ([vp, i], i),
        "tsd_debug_set_res_fuse_skip": ([vp, i], i),
        "tsd_debug_set_qkv_fuse": ([vp, i], i),
        "tsd_debug_mfma_sustained": ([vp, C.c_float, fp, fp], i),
        "tsd_debug_gemm_check": ([vp, i, i, i, i, i, i, i, i, i, i, fp, fp], i),
    }
    for name, (args, res) in sig.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = res


EXPORTS = None  # filled lazily for the symbol test


def declared_symbols():
    """Names of every function declared in include/tsd.h (parsed, so the test cannot drift)."""
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "tsd.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tsd_[a-z0-9_]+)\s*\(", txt)))


def last_error():
    return lib().tsd_last_error().decode()


def check(code, null_shape=None):
    """Map a status to the reference's behaviour.  Returns True when the caller should return a null matrix."""
    if code == TSD_OK:
        return False
    msg = last_error()
    if code == TSD_E_SHAPE and not _strict and null_shape is not None:
        print(msg + ". Returning null matrix")  # helpers/utils.mojo:1550-1552 convention
        return True
    raise TsdError(code, msg)


def f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a


def ptr(a):
    return None if a is None else a.ctypes.data_as(fp)


NULL_MATRIX = lambda: np.zeros((0, 0, 0), dtype=np.float32)  # noqa: E731  `Matrix(0,0,0)`

_default_ctx = None


class Context:
    """One per GPU: device, HIP stream, workspace arena (`tsd_ctx`)."""

    def __init__(self, device=0):
        h = vp()
        check(lib().tsd_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = device

    def synchronize(self):
        check(lib().tsd_ctx_synchronize(self.h))

    def timer_start(self):
        check(lib().tsd_ctx_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_float()
        check(lib().tsd_ctx_timer_stop(self.h, C.byref(ms)))
        return ms.value

    KERNEL_CLASSES = ("gemm", "conv3x3", "flash_attention", "groupnorm", "layernorm", "small_linear", "elementwise",
                      "softmax", "attn_tail_chain")

    def profile_begin(self):
        check(lib().tsd_ctx_profile_begin(self.h))

    def profile_end(self):
        """{class: (total_ms, launches)} measured with hipEvents on this context's stream."""
        nc = len(self.KERNEL_CLASSES)
        ms = (C.c_float * nc)()
        n = (C.c_int * nc)()
        check(lib().tsd_ctx_profile_end(self.h, ms, n, nc))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.KERNEL_CLASSES)}

    def profile_records(self, cap=8192):
        """[(class, M, N, K, batch, ms)] per launch of the current profiling pass."""
        rec = (C.c_int * (5 * cap))()
        ms = (C.c_float * cap)()
        n = lib().tsd_ctx_profile_records(self.h, rec, ms, cap)
        if n < 0:
            check(n)
        return [(self.KERNEL_CLASSES[rec[5 * i]], rec[5 * i + 1], rec[5 * i + 2], rec[5 * i + 3], rec[5 * i + 4], ms[i])
                for i in range(n)]

    def close(self):
        if self.h:
            lib().tsd_ctx_destroy(self.h)
            self.h = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(int(os.environ.get("LOCAL_RANK", "0")) if lib().tsd_device_count() > 1 else 0)
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx
