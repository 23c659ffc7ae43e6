"""The 4096 x 4096 d = 40 attention call (batch 8 x 8 heads): flash_attn_kernel<40, 2> (4-wave workgroups) against the 8-wave two-group
kernel of kernels_attn8.hip, interleaved in one process.  TSD_ATTN8_VAR selects a timing variant (-DTSD_ATTN8_VARIANTS builds);
with a -DTSD_ATTN8_TS build the mean ticks per wave in the X phase, at barrier 1, in the M phase and at barrier 2 are printed."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
import tsd
from tsd._lib import lib
L = lib()
ctx = tsd.default_context(); ms = C.c_float()
iters = int(os.environ.get("ITERS", 30)); rounds = int(os.environ.get("ROUNDS", 4))
shapes = [(40, 4096, 4096)] + ([(40, 4096, 77), (40, 1024, 1024)] if os.environ.get("ALL") else [])
for (d, S, Sk) in shapes:
    res = {2: [], 3: []}
    for r in range(rounds):
        for mode in (2, 3):
            L.tsd_debug_set_attn_qb(ctx.h, mode)
            rc = L.tsd_debug_attn_bench(ctx.h, 8, 8, d, S, Sk, iters, C.byref(ms))
            assert rc == 0, rc
            res[mode].append(ms.value * 1e3)
    fl = 4.0 * 64 * S * Sk * d
    for mode in (2, 3):
        v = sorted(res[mode]); med = v[len(v) // 2]
        print(f"d={d} Sq={S} Sk={Sk} mode={mode} var={os.environ.get('TSD_ATTN8_VAR', '0')}: median {med:7.1f} us  min {v[0]:7.1f}  "
              f"{fl / (med * 1e-6) / 1e12:6.1f} TF  ({fl / (med * 1e-6) / 2.5e15:.3f} of peak)  all={['%.1f' % x for x in res[mode]]}")
L.tsd_debug_set_attn_qb(ctx.h, 0)
try:
    f = L.tsd_debug_attn8_ticks   # -DTSD_ATTN8_TS builds only
except AttributeError:
    f = None
if f is not None:
    out = (C.c_double * 8)()
    L.tsd_debug_set_attn_qb(ctx.h, 3)
    L.tsd_debug_attn_bench(ctx.h, 8, 8, 40, 4096, 4096, 2, C.byref(ms))
    if f(out) == 0:
        for g in (0, 1):
            x, b1, m, b2 = [out[g * 4 + k] / 64.0 for k in range(4)]
            print(f"[ticks per tile] group {g}: X {x:7.1f}  barrier1 {b1:7.1f}  M {m:7.1f}  barrier2 {b2:7.1f}  sum {x + b1 + m + b2:7.1f}")
    L.tsd_debug_set_attn_qb(ctx.h, 0)
