"""ctypes face of oracle/cref/cref.c - the C restatement of the reference's CPU ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  `backend()` returns a namespace with the same function names and
signatures as `oracle.ops`, whose arithmetic runs in the C library; `oracle.models.using_ops(backend())` runs the module
graphs (UNet, Diffusion, VAE) on it.  Two uses: tests/test_cref_cpu.py holds the two statements of the algorithm (numpy
im2col + BLAS, and the reference's own loop nests in C) against each other and against tests/golden; bench.py's
`cpu_baseline` times the C loops - the shape of the reference's CPU path (`parallelize` over output channels / rows,
fp32) - on the host's cores.
"""
import ctypes
import os
import subprocess
import types

import numpy as np

from . import ops as _ops

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cref")
_SO = os.path.join(_DIR, "libcref.so")
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i, _l, _fl = ctypes.c_int, ctypes.c_long, ctypes.c_float
_SIG = {
    "cref_version": (ctypes.c_int, []),
    "cref_threads": (ctypes.c_int, []),
    "cref_set_threads": (None, [ctypes.c_int]),
    "cref_silu": (None, [_f, _f, _l]),
    "cref_gelu_tanh": (None, [_f, _f, _l]),
    "cref_quick_gelu": (None, [_f, _f, _l]),
    "cref_time_embedding": (None, [_fl, _f]),
    "cref_pad": (None, [_f, _i, _i, _i, _i, _i, _i, _i, _f]),
    "cref_conv2d": (None, [_f, _i, _i, _i, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _f]),
    "cref_upsample_nearest2x": (None, [_f, _i, _i, _i, _f]),
    "cref_group_norm": (None, [_f, _i, _l, _i, _fl, _f]),
    "cref_layer_norm": (None, [_f, _l, _i, _fl, _f]),
    "cref_matmul": (None, [_f, _f, _f, _i, _l, _l, _l, _i]),
    "cref_linear": (None, [_f, _f, _f, _f, _l, _l, _l]),
    "cref_softmax_rows": (None, [_f, _l, _l]),
    "cref_attention_core": (None, [_f, _f, _f, _f, _l, _l, _i, _i, _i]),
}


def build(force=False):
    """Compile oracle/cref/libcref.so with gcc (a few seconds).  Called by __graft_entry__.build() and, when the library
    is missing, by lib()."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_DIR, "cref.c")):
        subprocess.check_call(["make", "-C", _DIR, "-B", "libcref.so"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        for name, (res, args) in _SIG.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        assert L.cref_version() == 1
        _lib = L
    return _lib


def threads():
    return int(lib().cref_threads())


def threads_available():
    """CPUs this process may actually run on: the affinity mask, cut by the cgroup's CPU quota when there is one (a
    container that sees 256 cores may be allowed far fewer; an OpenMP team larger than that only spins)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = max(1, min(n, int(float(quota) / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_threads(n):
    lib().cref_set_threads(int(n))
    return threads()


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float32)


def _p(x):
    return None if x is None else x.ctypes.data_as(_f)


def silu(x):
    x = _a(x); y = np.empty_like(x); lib().cref_silu(_p(x), _p(y), x.size); return y


def gelu_tanh(x):
    x = _a(x); y = np.empty_like(x); lib().cref_gelu_tanh(_p(x), _p(y), x.size); return y


def quick_gelu(x):
    x = _a(x); y = np.empty_like(x); lib().cref_quick_gelu(_p(x), _p(y), x.size); return y


def time_embedding(t, sem=_ops.DEFAULT, dtype=np.float32):
    assert not sem.literal_time_freqs
    y = np.empty(320, np.float32); lib().cref_time_embedding(float(t), _p(y)); return y.astype(dtype)


def pad(x, pad_h=(0, 0), pad_w=(0, 0)):
    x = _a(x); C, H, W = x.shape
    y = np.empty((C, H + pad_h[0] + pad_h[1], W + pad_w[0] + pad_w[1]), np.float32)
    lib().cref_pad(_p(x), C, H, W, pad_h[0], pad_h[1], pad_w[0], pad_w[1], _p(y)); return y


def conv2d(x, w, b=None, padding=(0, 0), stride=(1, 1), pad_hw=None, max_cols_bytes=None):
    w = _a(w); O, I, k, k2 = w.shape
    assert k == k2
    x = _a(x[:I]); _, H, W = x.shape  # only the first in_channels are read (App.A D11)
    if pad_hw is None:
        pad_hw = ((padding[0], padding[0]), (padding[1], padding[1]))
    (pt, pb), (pl, pr) = pad_hw
    Ho = (H + pt + pb - k) // stride[0] + 1
    Wo = (W + pl + pr - k) // stride[1] + 1
    y = np.empty((O, Ho, Wo), np.float32)
    bb = None if b is None else _a(b)
    lib().cref_conv2d(_p(x), I, H, W, _p(w), _p(bb), O, k, pt, pb, pl, pr, stride[0], stride[1], _p(y)); return y


def upsample_nearest2x(x):
    x = _a(x); C, H, W = x.shape; y = np.empty((C, 2 * H, 2 * W), np.float32)
    lib().cref_upsample_nearest2x(_p(x), C, H, W, _p(y)); return y


def group_norm(x, num_groups, num_channels=None, eps=1e-5):
    if num_channels is None:
        num_channels = x.shape[0]
    assert num_channels <= x.shape[0] and num_channels % num_groups == 0
    xs = _a(x[:num_channels]); C, H, W = xs.shape; y = np.empty_like(xs)
    lib().cref_group_norm(_p(xs), C, H * W, num_groups, float(eps), _p(y)); return y


def layer_norm(x, eps=1e-5, sem=_ops.DEFAULT):
    assert not sem.literal_layernorm_global
    x = _a(x); y = np.empty_like(x)
    lib().cref_layer_norm(_p(x), int(np.prod(x.shape[:-1])), x.shape[-1], float(eps), _p(y)); return y


def linear(x, w, b=None):
    x = _a(x); w = _a(w); lead = x.shape[:-1]; K = x.shape[-1]; N = w.shape[0]
    assert w.shape[1] == K
    M = int(np.prod(lead)) if lead else 1
    y = np.empty((M, N), np.float32); bb = None if b is None else _a(b)
    lib().cref_linear(_p(x), _p(w), _p(bb), _p(y), M, K, N); return y.reshape(*lead, N)


def matmul(a, b):
    a = _a(a); b = _a(b)
    a3 = a.reshape((-1,) + a.shape[-2:]) if a.ndim > 2 else a[None]
    b3 = b.reshape((-1,) + b.shape[-2:]) if b.ndim > 2 else b[None]
    Bc, M, K = a3.shape
    assert b3.shape[1] == K and b3.shape[0] in (1, Bc)
    y = np.empty((Bc, M, b3.shape[2]), np.float32)
    lib().cref_matmul(_p(a3), _p(b3), _p(y), Bc, M, K, b3.shape[2], 1 if b3.shape[0] == 1 and Bc > 1 or b.ndim == 2 else 0)
    return y if a.ndim > 2 else y[0]


def softmax_lastdim(s, sem=_ops.DEFAULT):
    assert not sem.literal_softmax_axis
    y = np.array(s, dtype=np.float32, order="C")
    lib().cref_softmax_rows(_p(y), int(np.prod(y.shape[:-1])), y.shape[-1]); return y


def attention_core(q, k, v, H, causal=False, sem=_ops.DEFAULT):
    assert not sem.literal_head_split and not sem.literal_softmax_axis
    q, k, v = _a(q), _a(k), _a(v); Tq, D = q.shape; o = np.empty((Tq, D), np.float32)
    lib().cref_attention_core(_p(q), _p(k), _p(v), _p(o), Tq, k.shape[0], D, H, 1 if causal else 0); return o


def self_attention(x, H, w_in, b_in, w_out, b_out, causal=False, sem=_ops.DEFAULT):
    """`Self_Attention.forward` helpers/attention.mojo:26-65 on the C ops."""
    qkv = linear(x, w_in, b_in)
    q, k, v = np.split(qkv, 3, axis=-1)  # chunk(2, 3) on the feature axis (:29)
    return linear(attention_core(q, k, v, H, causal, sem), w_out, b_out)


def cross_attention(x, ctx, H, wq, bq, wk, bk, wv, bv, wo, bo, sem=_ops.DEFAULT):
    """`Cross_Attention.forward` helpers/attention.mojo:96-118 on the C ops."""
    q, k, v = linear(x, wq, bq), linear(ctx, wk, bk), linear(ctx, wv, bv)
    return linear(attention_core(q, k, v, H, False, sem), wo, bo)


_OVERRIDES = ("silu", "gelu_tanh", "quick_gelu", "time_embedding", "pad", "conv2d", "upsample_nearest2x", "group_norm",
              "layer_norm", "linear", "matmul", "softmax_lastdim", "attention_core", "self_attention", "cross_attention")


def backend():
    """A namespace with oracle.ops' names: the arithmetic above in C, everything else (concat, layout helpers, the
    extension norms) from oracle.ops."""
    lib()
    ns = types.SimpleNamespace(**{k: v for k, v in vars(_ops).items() if not k.startswith("__")})
    for name in _OVERRIDES:
        setattr(ns, name, globals()[name])
    return ns
