#!/usr/bin/env python3
"""bench.py - headline benchmark of the Tiny-SD hot path on MI355X.

Metric (BASELINE.json): UNet denoising steps/sec @ 512x512 (latent 4x64x64), batch = 8 per GPU.
One "step" = one batched `Diffusion.forward` over the rank's 8 samples + the DDPM update
(pipeline.mojo:87-122), inputs resident in HBM, fp16 storage / fp32 accumulate, random-init weights,
synthetic latents/context (there is no network for checkpoints or datasets).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU; the batch of independent prompts is sharded (8 per rank, weak scaling), the
only collective is the one-time RCCL broadcast of the packed weights from rank 0 (outside the timed
region).  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` started PLAINLY (no WORLD_SIZE in the environment) launches its own N ranks: it re-executes
itself under `python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node N`, one rank per visible
device (started under mpirun / srun without torchrun's variables it refuses instead of starting N ranks per rank).  The line never lies about N: fewer than N visible devices, or a WORLD_SIZE that differs from --gpus, is an error
(exit status 2), and `n_gpus` is asserted equal to --gpus before anything is printed; `rccl_ranks` reports the size of the
communicator that actually ran (torch.distributed's world size, and ncclCommCount on the library's own RCCL path).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0  # MI355X dense fp16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
SEED = 1234


def shard_range(total, rank, world):
    """Contiguous batch shards: sample i -> rank floor(i*world/total)."""
    lo = (total * rank) // world
    hi = (total * (rank + 1)) // world
    return lo, hi


class _DevView:
    """Zero-copy torch view of a device allocation (the packed weight blob) through __cuda_array_interface__."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(models, rank, world, device, backend="nccl"):
    """One broadcast per model of the packed weight blob from rank 0.

    backend "nccl" (= RCCL over xGMI on ROCm): `dist.broadcast` runs directly ON the blob - a zero-copy torch view of the
    library's device allocation, no staging tensor.  backend "gloo" (CPU control plane: the N > 1 code path on a one-GPU
    box, and the CPU tests, where `device` is "cpu" and the "blob" is host memory): the bytes go through a host tensor.
    Any failure raises - with more than one rank the caller must NOT fall back to per-rank initialisation silently."""
    import torch
    import torch.distributed as dist
    if os.environ.get("TSD_BENCH_FAIL_BCAST") == "1":
        raise RuntimeError("weight broadcast failure injected by TSD_BENCH_FAIL_BCAST=1")
    t0 = time.time()
    nbytes = 0
    for m in models:
        ptr, n = m.packed_blob()
        if device == "cpu":  # host "blob" (tests): a writable view of the caller's buffer
            buf = torch.frombuffer((ctypes.c_uint8 * n).from_address(ptr), dtype=torch.uint8)
            dist.broadcast(buf, 0)
        elif backend == "nccl":
            m.ctx.synchronize()
            buf = torch.as_tensor(_DevView(ptr, n), device=device)
            assert buf.data_ptr() == ptr, "torch copied the blob instead of aliasing it"
            dist.broadcast(buf, 0)
            torch.cuda.synchronize()
        else:  # gloo: device blob -> pinned host tensor -> gloo broadcast -> device blob
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            m.ctx.synchronize()
            host = torch.empty(n, dtype=torch.uint8)
            if rank == 0:
                assert hip.hipMemcpy(host.data_ptr(), ptr, n, 2) == 0  # hipMemcpyDeviceToHost
            dist.broadcast(host, 0)
            if rank != 0:
                assert hip.hipMemcpy(ptr, host.data_ptr(), n, 1) == 0  # hipMemcpyHostToDevice
        if rank != 0:
            m.mark_loaded()
        nbytes += n
    return time.time() - t0, nbytes


def exchange_unique_id(rank, make_id):
    """Rank 0's 128-byte RCCL unique id to every rank through the existing torch.distributed control plane (whatever its
    backend: the id is a host byte string).  `make_id()` is called on rank 0 only."""
    import torch.distributed as dist
    box = [make_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = bytes(box[0])
    assert len(uid) == 128, len(uid)
    return uid


def broadcast_weights_native(models, ctx, rank, world):
    """TSD_BENCH_NATIVE_DIST=1: the library's OWN RCCL path (`tsd_dist_init` / `tsd_dist_broadcast_weights`, include/tsd.h) -
    what a Mojo host without torch would call; torch.distributed only carries the 128-byte unique id.  Failures raise."""
    from tsd._lib import check, lib
    t0 = time.time()
    uid = exchange_unique_id(rank, lambda: _native_unique_id())
    buf = ctypes.create_string_buffer(uid, 128)
    check(lib().tsd_dist_init(ctx.h, rank, world, buf))
    nbytes = 0
    for m in models:
        check(lib().tsd_dist_broadcast_weights(m.h, 0))
        if rank != 0:
            m.mark_loaded()
        nbytes += m.packed_blob()[1]
    ctx.synchronize()
    cnt = ctypes.c_int(0)
    check(lib().tsd_dist_comm_count(ctx.h, ctypes.byref(cnt)))
    assert cnt.value == world, f"ncclCommCount says {cnt.value} ranks, the launcher {world}"
    check(lib().tsd_dist_finalize(ctx.h))
    return time.time() - t0, nbytes


def _native_unique_id():
    from tsd._lib import check, lib
    uid = ctypes.create_string_buffer(128)
    check(lib().tsd_dist_unique_id(uid))
    return uid.raw


def pmc_traffic_per_gemm_launch():
    """(HBM bytes per gemm_kernel launch, source file) from the newest committed PMC profile (rocprofv3 --pmc FETCH_SIZE /
    WRITE_SIZE in separate passes over this same command, scripts/gpu_pmc_bench.sh -> profiles/rNN_pmc_hbm_traffic.txt),
    with the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of wide coalesced reads -> x2).
    (None, None) when no profile is committed: bench.py itself never runs a profiler, so the figure is NOT measured by
    this run - `traffic_source` in the JSON line says which file it came from."""
    for name in ("r06_pmc_hbm_traffic.txt", "r05_pmc_hbm_traffic.txt", "r04_pmc_hbm_traffic.txt", "r03_pmc_hbm_traffic.txt", "r02_pmc_hbm_traffic.txt", "r01_pmc_hbm_traffic.txt"):
        path = os.path.join(ROOT, "profiles", name)
        try:
            fetch = write = launches = 0.0
            for line in open(path).read().splitlines()[1:]:
                parts = line.split()
                if "gemm_kernel" not in line or len(parts) < 4:
                    continue
                n, f, w = float(parts[-3]), float(parts[-2]), float(parts[-1])  # dispatches, KiB/dispatch, KiB/dispatch
                fetch += n * f; write += n * w; launches += n
            if launches:
                return int((2.0 * fetch + write) * 1024 / launches), "profiles/" + name
        except OSError:
            continue
    return None, None


def flash_workgroups_per_step(B, L):
    """Workgroups of the nine self-attention launches of one UNet step - the denominator of the exact-repeat fraction.  d = 40: 512 queries
    per workgroup on the 8-wave kernel (kernels_attn8.hip: key loops of >= 512 keys with >= 64 workgroups per sample, i.e. L = 64), else
    4 waves of 64; d = 80 / 160: 4 waves of 32 queries; 8 heads."""
    n, side = 0, L
    for level, q_per_wg in enumerate((256, 128, 128)):
        S = side * side
        if level == 0 and os.environ.get("TSD_ATTN_WG8", "1") != "0" and S >= 512 and -(-S // 512) * 8 >= 64:
            q_per_wg = 512
        n += 3 * B * 8 * max(1, -(-S // q_per_wg))
        side //= 2
    return n


def peaked_logits_bench(tsd, ctx, B, L, T, lat, cx, noise, n_sched, K, friendly_steps_per_s):
    """The headline loop under attention statistics that are NOT friendly to the flash kernel's optimistic softmax pass.

    The kernel fixes its softmax reference after key tile 0 (plus the query's own key block) and repeats a workgroup with the exact,
    per-tile-maximum pass only when a later score overflows fp16 (kernels_attn.hip).  With random-init weights no workgroup repeats,
    so the headline number times the fast path only; trained attention is far more peaked.  Here the in_proj weights of all nine
    self-attention layers are scaled by s (scores by s^2) on a second copy of the model, s chosen so that about 10 % of the flash
    workgroups repeat, and so that all of them do; the same 50-step loop is timed for both.  The fused tail's 77-key cross attention
    subtracts the exact row maximum (no optimistic pass): its time does not depend on the data."""
    from tsd._lib import TsdError, lib
    wg = flash_workgroups_per_step(B, L)
    pk = tsd.Diffusion(seed=SEED, ctx=ctx)
    kind = pk.model.kind
    idx = [i for i, (name, shape, used, bound) in enumerate(pk.model.specs) if used and name.endswith("layer4.in_proj.weight")]
    assert len(idx) == 9, [pk.model.specs[i][0] for i in idx]
    base = {i: tsd.rng.uniform(SEED, kind * 4096 + i, int(np.prod(pk.model.specs[i][1])), pk.model.specs[i][3]).reshape(pk.model.specs[i][1])
            for i in idx}
    sess = tsd.Session(pk.model, None, B, L, T, cfg=False)
    sess.set_schedule(1000, n_sched, 0)

    def run(scale, steps, warm):
        for i in idx:
            pk.model.set_param(i, base[i] * np.float32(scale))
        sess.upload(lat, cx, None, noise)
        for i in range(warm):
            sess.step(i)
        ctx.synchronize()
        lib().tsd_debug_attn_exact_passes(ctx.h, 1)
        ctx.timer_start()
        for i in range(steps):
            sess.step((warm + i) % n_sched)
        ms = ctx.timer_stop()
        n = lib().tsd_debug_attn_exact_passes(ctx.h, 1)
        ctx.synchronize()  # raises TSD_E_NONFINITE if the scaled model overflowed fp16
        return n / float(steps * wg), ms / steps

    probe = {}
    for s_ in (1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0, 5.0, 6.0, 8.0):
        try:
            probe[s_] = run(s_, 3, 1)[0]
        except TsdError as e:
            probe[s_] = None
            lib().tsd_debug_nonfinite_count(ctx.h, 1)
            break
        if probe[s_] >= 0.995:
            break
    usable = {k: v for k, v in probe.items() if v is not None}
    out = {"flash_workgroups_per_step": wg, "friendly_steps_per_s": round(friendly_steps_per_s, 3),
           "exact_repeat_fraction_by_in_proj_scale": {str(k): (None if v is None else round(v, 4)) for k, v in probe.items()},
           "what": "in_proj weights of the nine self-attention layers scaled by s (attention scores by s^2) on a copy of the model; fraction "
                   "= flash-attention workgroups that repeated their softmax exactly / all flash workgroups, over the timed steps; "
                   "the fused tail's cross attention has no optimistic pass (data-independent)"}
    picks = {}
    some = {k: v for k, v in usable.items() if 0.0 < v < 0.995}
    if some:
        picks["about_10pct"] = min(some, key=lambda k: abs(some[k] - 0.10))
    if usable:
        picks["all"] = max(usable, key=lambda k: (usable[k], -k))
    for name, s_ in picks.items():
        try:
            frac, ms = run(s_, K, 2)
            out[name] = {"in_proj_scale": s_, "exact_repeat_fraction": round(frac, 4), "ms_per_step": round(ms, 4),
                         "steps_per_s": round(1e3 / ms, 3), "vs_friendly": round(1e3 / ms / friendly_steps_per_s, 4)}
        except TsdError as e:
            out[name] = {"in_proj_scale": s_, "error": str(e)}
            lib().tsd_debug_nonfinite_count(ctx.h, 1)
    sess.close()
    pk.model.close()
    return out


# one tile per CU in every run: rows x columns of the problem = 256 tiles of the configuration
K_LOOP_PROBES = (
    # (label, conv, B, H, W, N, Cin of the in-step launch (K lengths = 1x..4x), tile configuration, FM, FN, compute waves per SIMD, tile rows, tile columns)
    ("conv3x3 C->320 @64x64, 256x160 staggered tile + loader waves (cfg 51: the six 64x64-level convs)", 1, 8, 64, 64, 320, 320, 51, 4, 5, 2, 256, 160),
    ("conv3x3 C->640 @32x32, 128x160 tile, 3-slot ring (cfg 5: the 32x32-level convs)", 1, 8, 32, 32, 640, 640, 5, 4, 5, 1, 128, 160),
    ("dense 8192x640xK, 128x160 staggered tile + loader waves (cfg 54: 15 launches per step at K = 640)", 0, 8, 32, 32, 640, 640, 54, 2, 5, 2, 128, 160),
    ("dense 2048x1280xK, 64x160 tile + loader waves (cfg 47: 15 launches per step at K = 1280)", 0, 8, 16, 16, 1280, 1280, 47, 2, 5, 1, 64, 160),
)
K_LOOP_LENGTHS, K_LOOP_REPEATS, K_LOOP_ITERS, K_LOOP_FIXED_TOL_US = (1, 2, 3, 4), 3, 20, 2.0


def k_loop_model(tsd, dev_index, in_step=None, probes=K_LOOP_PROBES):
    """What a K tile and a launch cost, as a MEASUREMENT: the same one-tile-per-CU launch (real epilogue: bias + residual) at FOUR K
    lengths, three interleaved repeats, a least-squares line per repeat: time = fixed + slope x K-tiles.  Reported: mean slope and
    intercept, their spread over the repeats (half the range) and the residual of the fit.  A configuration whose intercept moves by
    more than +-2 us between repeats is reported as `unstable` with its raw times and NO derived figure (round 5 printed a two-point
    intercept that swung 9x between boxes).  slope -> the chip-wide rate INSIDE the K loop against the fp16 MFMA peak, next to the MFMA
    clocks the tile needs (16 per v_mfma_f32_16x16x32_f16, per SIMD); fixed -> prologue, ring fill, drain and epilogue of a REPEATED,
    WARM launch.  `in_step` = the per-launch records of the profiled step (class, M, N, K, batch, ms): the same shape's duration inside
    the step, and the difference to the warm model at that K (`cold_us_per_launch`: operands that are not in the L2 when the launch
    starts - what the model does not contain)."""
    import ctypes as C
    from tsd._lib import lib
    old = os.environ.get("TSD_BENCH_EPI")
    os.environ["TSD_BENCH_EPI"] = "1"   # read once by tsd_ctx_create: the probe launches carry the projection / conv2 epilogue
    try:
        c2 = tsd.Context(dev_index)
    finally:
        if old is None:
            os.environ.pop("TSD_BENCH_EPI", None)
        else:
            os.environ["TSD_BENCH_EPI"] = old
    out = []
    for label, conv, B_, H, W, N, cin0, cfg, FM, FN, wps, BM, BN in probes:
        cins = [cin0 * m for m in K_LOOP_LENGTHS]
        kts = [(9 * c if conv else c) // 64 for c in cins]
        us = np.zeros((K_LOOP_REPEATS, len(cins)))
        ok = True
        for rep in range(K_LOOP_REPEATS):
            for i, cin in enumerate(cins):
                ms = C.c_float()
                if lib().tsd_debug_gemm_bench(c2.h, conv, B_, H, W, cin, N, 1, 0, cfg, K_LOOP_ITERS, C.byref(ms)) != 0:
                    ok = False
                    break
                us[rep, i] = ms.value * 1e3
            if not ok:
                break
        if not ok:
            out.append({"tile": label, "error": "tsd_debug_gemm_bench failed"})
            continue
        fits = [np.polyfit(kts, us[rep], 1) for rep in range(K_LOOP_REPEATS)]   # (slope, intercept) per repeat
        slopes, fixeds = np.array([f[0] for f in fits]), np.array([f[1] for f in fits])
        slope, fixed = float(slopes.mean()), float(fixeds.mean())
        slope_spread, fixed_spread = float(np.ptp(slopes) / 2), float(np.ptp(fixeds) / 2)
        resid = float(np.sqrt(np.mean([(np.polyval(f, kts) - us[rep]) ** 2 for rep, f in enumerate(fits)])))
        rec = {"tile": label, "cfg": cfg, "k_tiles": kts, "launch_us": [[round(float(u), 2) for u in row] for row in us],
               "us_per_k_tile": round(slope, 4), "us_per_k_tile_spread": round(slope_spread, 4),
               "fixed_us_per_launch": round(fixed, 2), "fixed_us_spread": round(fixed_spread, 2), "fit_residual_us_rms": round(resid, 3),
               "fit": "least squares over %d K lengths, %d repeats of %d launches; spread = half the range over the repeats" % (len(kts), K_LOOP_REPEATS, K_LOOP_ITERS)}
        if fixed_spread > K_LOOP_FIXED_TOL_US or slope <= 0:
            rec["unstable"] = True   # no derived figure from an intercept that does not reproduce on this box
            out.append(rec)
            continue
        flop_kt = 256.0 * BM * BN * 64 * 2          # all 256 CUs, one K tile each
        loop_tf = flop_kt / (slope * 1e-6) / 1e12
        mfma_clk = wps * 2 * FM * FN * 16
        meas_clk = slope * 1e-6 * 2.4e9             # clocks of the 2.4 GHz the nominal peak assumes
        rec.update({"mfma_clk_per_k_tile": mfma_clk, "measured_clk_per_k_tile_at_2p4GHz": round(meas_clk, 0),
                    "lds_and_barrier_clk_per_k_tile": round(meas_clk - mfma_clk, 0),
                    "k_loop_tflops": round(loop_tf, 1), "k_loop_frac_of_peak": round(loop_tf / PEAK_FP16_TFLOPS, 4),
                    "whole_launch_frac_of_peak": {str(kt): round(flop_kt * kt / ((fixed + slope * kt) * 1e-6) / 1e12 / PEAK_FP16_TFLOPS, 4)
                                                   for kt in kts}})
        if in_step:
            rows, K0 = B_ * H * W, (9 * cin0 if conv else cin0)
            hit = [r[5] * 1e3 for r in in_step if r[0] == ("conv3x3" if conv else "gemm") and r[2] == N and r[3] == K0
                   and (r[1] == rows or r[1] * max(1, r[4]) == rows)]
            if hit:
                warm = fixed + slope * kts[0]
                rec.update({"in_step_launches": len(hit), "in_step_us_per_launch": round(float(np.mean(hit)), 2),
                            "warm_model_us_at_that_k": round(warm, 2), "cold_us_per_launch": round(float(np.mean(hit)) - warm, 2),
                            "in_step_note": "in-step times: hipEvent pairs of the profiling pass, roofline.event_overhead_us_per_launch_removed taken off"})
        out.append(rec)
    lib().tsd_ctx_destroy(c2.h)
    c2.h = None
    return out


def cpu_baseline(L, T):
    """The CPU restatements of the reference's algorithm ('port') timed on this box's host cores on a bounded sample: ONE
    sample-step (1/8 of a batch-8 step: one Diffusion.forward + DDPM update at L=64).  Two statements are timed: the
    reference's own loop nests in C with OpenMP where the reference calls `parallelize` (oracle/cref/cref.c - the shape of the
    reference's CPU path, and the reported value) and the numpy oracle (im2col + BLAS).  Returns (t_c per sample-step, threads, t_numpy, repetitions of the C leg)."""
    from oracle import cref, models as omodels, ops as oops, rng as orng, sampler as osampler, spec as ospec
    P = ospec.init_params("diffusion", SEED, only_used=True)
    lat = orng.normal(SEED, 2, 4 * L * L).reshape(4, L, L)
    ctx = orng.normal(SEED, 5, T * 768).reshape(T, 768)
    noise = orng.normal(SEED, 3, 4 * L * L).reshape(4, L, L)
    s = osampler.DDPMSampler(1000)
    s.set_inference_timesteps(50)

    def sample_step(o):
        t0 = time.time()
        eps = omodels.diffusion(P, lat, ctx, o.time_embedding(980.0))
        x = s.step(980, lat, eps, noise)
        dt = time.time() - t0
        assert np.isfinite(x).all()
        return dt, x

    n_thr = cref.threads_available()  # the CPUs this process may run on (cgroup quota), not the host's count
    try:
        cb = cref.backend()
    except Exception as e:  # noqa: BLE001 - no gcc / OpenMP on this host: the numpy statement alone (returned t_c is None)
        print(f"bench.py: oracle/cref/libcref.so unavailable ({e}); cpu_baseline falls back to the numpy oracle", file=sys.stderr)
        cb = None
    t_c, reps, x_c = None, 0, None
    if cb is not None:
        n_thr = cref.set_threads(n_thr)
        t_c = 0.0
        with omodels.using_ops(cb):
            while reps < 8 and t_c < 10.0:  # bounded sample: at most one batch worth of sample-steps, about 10 s of CPU work
                dt, x_c = sample_step(cb)
                t_c += dt
                reps += 1
        t_c /= reps
    try:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=n_thr):
            t_np, x_np = sample_step(oops)
    except ImportError:
        t_np, x_np = sample_step(oops)
    if x_c is not None:
        assert np.linalg.norm(x_c - x_np) <= 1e-4 * np.linalg.norm(x_np)  # the two statements agree on the timed sample
    return t_c, n_thr, t_np, reps


def visible_devices():
    """HIP devices this process can see (through libtsd: the product's own view, HIP_VISIBLE_DEVICES applied)."""
    from tsd._lib import lib
    return int(lib().tsd_device_count())


def self_launch(n):
    """`bench.py --gpus N` without a launcher: start N ranks of this same command line under torch.distributed.run, one per visible
    device (TSD_BENCH_DEVICE / TSD_BENCH_BACKEND=gloo - the one-GPU exercise of the N > 1 path - lift the device-count check).
    Returns the exit status to leave with."""
    import subprocess
    one_device = "TSD_BENCH_DEVICE" in os.environ or os.environ.get("TSD_BENCH_LAUNCH_ONLY") == "1"
    if not one_device:
        have = visible_devices()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} HIP device(s) visible: refusing to run", file=sys.stderr)
            return 2
    # --standalone: the launcher's own c10d store picks the rendezvous port (no bind / close / reuse race on a busy host: ADVICE r05);
    # --local-addr: the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    print(f"bench.py: --gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd[1:7])} ...", file=sys.stderr)
    return subprocess.call(cmd, env=env)


def launch_only(rank, world, args):
    """TSD_BENCH_LAUNCH_ONLY=1: rendezvous, count the ranks, print the launcher's half of the line - no GPU touched.  How the CPU
    test-suite checks that `bench.py --gpus 2`, started plainly, really is two ranks."""
    import torch
    import torch.distributed as dist
    dist.init_process_group(os.environ.get("TSD_BENCH_BACKEND", "gloo"))
    t = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(t)
    ranks = int(t.item())
    ok = ranks == world == args.gpus == dist.get_world_size()
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "rccl_ranks": {"torch_distributed": dist.get_world_size(), "counted": ranks},
                          "shards": [list(shard_range(world * args.batch, r, world)) for r in range(world)]}), flush=True)
    dist.destroy_process_group()
    return 0 if ok else 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="samples per GPU (BASELINE config 2: 8)")
    ap.add_argument("--latent", type=int, default=64, help="latent side (64 = 512x512)")
    ap.add_argument("--tokens", type=int, default=77)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the VAE-decode timing used for images/s")
    # the other BASELINE configs ride along as extra fields (a few seconds in total); --no-extras / --no-<name> skip them
    ap.add_argument("--no-cfg", action="store_true", help="skip the classifier-free-guidance loop (UNet batch 2B per step)")
    ap.add_argument("--no-img2img", action="store_true",
                    help="skip BASELINE configs[3] (VAE encoder + 30 UNet steps + VAE decoder)")
    ap.add_argument("--no-sd15", action="store_true", help="skip BASELINE configs[4] (full-size 860 M parameter UNet, batch 4)")
    ap.add_argument("--no-peaked", action="store_true", help="skip the headline loop under peaked attention logits (exact-repeat softmax path)")
    ap.add_argument("--no-kloop", action="store_true", help="skip the live K-loop model of the GEMM tiles (roofline.k_loop_model)")
    ap.add_argument("--no-extras", action="store_true", help="headline only: same as --no-cfg --no-img2img --no-sd15 --no-peaked --no-kloop")
    ap.add_argument("--cfg", action="store_true", help=argparse.SUPPRESS)      # accepted for compatibility: now on by default
    ap.add_argument("--img2img", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--sd15", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    args.cfg = not (args.no_cfg or args.no_extras)
    args.img2img = not (args.no_img2img or args.no_extras or args.no_decode)
    args.sd15 = not (args.no_sd15 or args.no_extras)
    args.peaked = not (args.no_peaked or args.no_extras)
    args.kloop = not (args.no_kloop or args.no_extras)

    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    # started by another launcher that does not export torchrun's variables (mpirun, srun): every rank would start N more - refuse
    foreign = [v for v in ("OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "SLURM_NTASKS") if int(os.environ.get(v, "1") or 1) > 1]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if foreign:
            print(f"bench.py: --gpus {args.gpus} under a launcher that sets {foreign[0]} but not RANK / WORLD_SIZE / MASTER_*: start it with "
                  "torch.distributed.run (or plainly, one process) - refusing to start ranks per rank", file=sys.stderr)
            sys.exit(2)
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # a line that says n_gpus = 1 for a run asked to measure 8 (or the reverse) is worse than no line
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to run (launch with --nproc-per-node {args.gpus}, "
              "or plainly and let bench.py start its own ranks)", file=sys.stderr)
        sys.exit(2)
    B, L, T = args.batch, args.latent, args.tokens
    if os.environ.get("TSD_BENCH_LAUNCH_ONLY") == "1":
        sys.exit(launch_only(rank, world, args))

    os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")  # before anything initialises HIP (see tsd/_lib.py)
    if world > 1 or os.environ.get("TSD_BENCH_FORCE_DIST") == "1":
        import torch  # noqa: F401  BEFORE libtsd: a process must hold ONE HIP runtime, and torch bundles its own
    import tsd
    tsd.set_strict(True)
    dist = None
    # TSD_BENCH_DEVICE / TSD_BENCH_BACKEND exist only to exercise the N>1 code path on a one-GPU box (all ranks on one
    # device, gloo for the control collectives); the real run is one rank per GPU over RCCL.
    dev_index = int(os.environ.get("TSD_BENCH_DEVICE", local_rank))
    backend = os.environ.get("TSD_BENCH_BACKEND", "nccl")
    # TSD_BENCH_FORCE_DIST=1 takes the torch.distributed path (RCCL init, blob broadcast, barrier, MAX-reduce) even with
    # one rank, so the coexistence of torch's RCCL and libtsd in one process can be exercised on a one-GPU box.
    use_dist = world > 1 or os.environ.get("TSD_BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{dev_index}"))
        else:
            dist.init_process_group(backend)
    ctx = tsd.Context(dev_index)
    tsd.set_default_context(ctx)

    # weights: rank 0 random-initialises on the device; other ranks receive the packed blob over RCCL
    unet = tsd.Diffusion(seed=SEED if rank == 0 else None, ctx=ctx)
    dec = None if args.no_decode else tsd.Decoder(seed=SEED if rank == 0 else None, ctx=ctx)
    bcast_s, bcast_bytes, bcast_how = 0.0, 0, "none (single GPU)"
    rccl_ranks = {"torch_distributed": dist.get_world_size() if use_dist else 1, "backend": backend if use_dist else None,
                  "ncclCommCount_native": None}
    if use_dist:
        models = [unet.model] + ([dec.model] if dec is not None else [])
        # a failed broadcast is FATAL: a multi-GPU run that silently re-initialised every rank from the seed would hide
        # exactly the failure the run exists to detect
        if os.environ.get("TSD_BENCH_NATIVE_DIST") == "1":
            bcast_s, bcast_bytes = broadcast_weights_native(models, ctx, rank, world)
            rccl_ranks["ncclCommCount_native"] = world  # asserted inside
            bcast_how = "rccl broadcast of the packed blobs from rank 0 by libtsd itself (tsd_dist_*; unique id over torch.distributed)"
        else:
            bcast_s, bcast_bytes = broadcast_weights(models, rank, world, f"cuda:{dev_index}", backend)
            bcast_how = ("rccl broadcast of the packed blobs from rank 0 (in place, zero-copy view of the blob)" if backend == "nccl"
                         else f"{backend} broadcast of the packed blobs from rank 0 through host memory")

    # derived device buffers (K-tile-major weight copies, fused-kernel weight streams): rebuilt on EVERY rank from the packed
    # weights after the broadcast - timed here so a multi-GPU run shows what start-up costs beside the broadcast itself
    t_d = time.time()
    unet.model.prepare()
    if dec is not None:
        dec.model.prepare()
    derived_s = time.time() - t_d

    # synthetic inputs: global batch of world*B independent prompts, this rank's contiguous shard
    lo, hi = shard_range(world * B, rank, world)
    assert hi - lo == B
    n_sched = 50
    nl = 4 * L * L
    lat = np.stack([tsd.rng.normal(SEED, 1000 + i, nl).reshape(4, L, L) for i in range(lo, hi)])
    cx = np.stack([tsd.rng.normal(SEED, 2000 + i, T * 768).reshape(T, 768) for i in range(lo, hi)])
    noise = tsd.rng.normal(SEED, 3000 + rank, n_sched * B * nl).reshape(n_sched, B, 4, L, L)

    sess = tsd.Session(unet.model, dec.model if dec is not None else None, B, L, T, cfg=False)
    sess.set_schedule(1000, n_sched, 0)  # 50 DDPM steps: t = 980, 960, ..., 0
    sess.upload(lat, cx, None, noise)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()

    K, W = args.steps, args.warmup
    for i in range(W):
        sess.step(i % n_sched)
    barrier()
    from tsd._lib import lib as _tsd_lib
    _tsd_lib().tsd_debug_attn_exact_passes(ctx.h, 1)
    t0 = time.perf_counter()
    ctx.timer_start()
    for i in range(K):
        sess.step((W + i) % n_sched)
    ev_ms = ctx.timer_stop()  # hipEvents on the library's own stream (synchronises the stream)
    barrier()
    dt = time.perf_counter() - t0
    # flash-attention workgroups that had to repeat their softmax exactly inside the timed region (kernels_attn.hip)
    attn_exact_wg = _tsd_lib().tsd_debug_attn_exact_passes(ctx.h, 1)
    rank_rate = (K / (ev_ms * 1e-3), K / (ev_ms * 1e-3))  # (min, max) over ranks of this rank's own steps/s (device events)
    if dist is not None:
        import torch
        dd = f"cuda:{dev_index}" if backend == "nccl" else "cpu"
        tt = torch.tensor([dt], dtype=torch.float64, device=dd)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # stragglers: every rank's own rate from its device events, min and max over ranks (a real 8-GPU run shows them here)
        rr = torch.tensor([K / (ev_ms * 1e-3), -K / (ev_ms * 1e-3), derived_s], dtype=torch.float64, device=dd)
        dist.all_reduce(rr, op=dist.ReduceOp.MAX)
        rank_rate = (-float(rr[1].item()), float(rr[0].item()))
        derived_s = float(rr[2].item())  # the slowest rank's rebuild
    out_lat = sess.latents()
    finite = bool(np.isfinite(out_lat).all())

    if rank == 0:
        # ---- per-kernel-class pass (hipEvent pair around every launch, same stream) -> roofline of the dominant kernel
        P = max(2, min(5, K))
        ctx.profile_begin()
        for i in range(P):
            sess.step((W + i) % n_sched)
        recs = ctx.profile_records()  # per launch: (class, M, N, K, batch, ms) - read before profile_end() clears them
        prof = ctx.profile_end()
        per_step = {k: (ms / P, n // P) for k, (ms, n) in prof.items()}
        # A hipEvent pair brackets a launch PLUS its dispatch gaps; summed over the ~190 launches of a step the bracketed
        # times exceed the step measured without events.  That excess, spread evenly per launch, is taken off every
        # record so the per-class sums add up to the un-instrumented step (and the GEMM average agrees with
        # `rocprofv3 --kernel-trace --stats`, which times kernels only: 36.6 us there vs 39.2 us uncorrected).
        n_launch = sum(v[1] for v in per_step.values())
        ev_overhead_ms = max(0.0, (sum(v[0] for v in per_step.values()) - 1e3 * dt / K) / max(1, n_launch))
        per_step = {k: (max(0.0, ms - n * ev_overhead_ms), n) for k, (ms, n) in per_step.items()}
        gemm_ms = per_step["gemm"][0] + per_step["conv3x3"][0]
        gemm_launches = per_step["gemm"][1] + per_step["conv3x3"][1]
        # algorithmic FLOP of the GEMM-class launches of one step = conv + linear share of 408.33 GFLOP/sample
        # (conv 195.22 + linear 137.51 at L=64; attention core is its own kernel)
        total_gf = tsd.flop_count("diffusion", L, T)
        attn_gf = 0.0
        side = L
        for hd, cnt in ((40, 3), (80, 3), (160, 3)):
            S = side * side
            attn_gf += cnt * (2 * 2 * 8 * S * S * hd + 2 * 2 * 8 * S * T * hd) / 1e9
            side //= 2
        # the fused attention-tail kernel (kernels_chain.hip) does the six row-local GEMMs of the 64x64-level attention blocks
        # (32 C^2 flop per token row) and their cross-attention core (4 T C per row) itself: that work is taken OFF the
        # gemm_kernel's account and reported under the tail kernel's own roofline entry below
        # (the fused HEAD kernel of the same blocks - GroupNorm-apply, conv_in, LayerNorm, q/k/V^T projections: 8 C^2 per
        # row - is recorded in the same class with K = 1)
        chain_recs = [r for r in recs if r[0] == "attn_tail_chain"]
        chain_lin_gf = sum(r[1] * (8.0 if r[3] == 1 else 32.0) * r[2] * r[2] for r in chain_recs) / P / 1e9
        chain_att_gf = sum(r[1] * 4.0 * T * r[2] for r in chain_recs if r[3] == 0) / P / 1e9
        gemm_gf_step = (total_gf - attn_gf) * B - chain_lin_gf
        achieved = (gemm_gf_step / 1e3) / (gemm_ms / 1e3) if gemm_ms > 0 else 0.0  # TFLOP/s
        traffic, traffic_src = pmc_traffic_per_gemm_launch()
        roofline = {"bound": "mfma", "kernel": "gemm_kernel<...> (dense + conv3x3 implicit GEMM)",
                    "achieved": round(achieved, 2), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_FP16_TFLOPS, 4), "traffic": traffic,
                    "traffic_source": None if traffic is None else f"{traffic_src} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                      "passes over this same command, scripts/gpu_pmc_bench.sh; committed file, NOT measured by this run)",
                    "launches_per_step": gemm_launches, "avg_launch_us": round(1e3 * gemm_ms / max(1, gemm_launches), 2),
                    "algorithmic_gflop_per_step": round(gemm_gf_step, 1),
                    "event_overhead_us_per_launch_removed": round(1e3 * ev_overhead_ms, 3),
                    "per_class_ms_per_step": {k: round(v[0], 4) for k, v in per_step.items()},
                    "per_class_launches_per_step": {k: v[1] for k, v in per_step.items()}}
        # secondary kernels against their own rooflines (same pass, same per-launch correction): attention on the fp16
        # MFMA peak (algorithmic 4*B*H*Sq*Sk*d per launch), the norms on HBM (algorithmic bytes = read once + write once)
        sec = {}
        for cls in ("flash_attention", "attn_tail_chain", "groupnorm", "layernorm"):
            ms_c, n_c = per_step[cls]
            if ms_c <= 0:
                continue
            rs = [r for r in recs if r[0] == cls]
            if cls == "attn_tail_chain":
                tf = (chain_lin_gf + chain_att_gf) / 1e3 / (ms_c / 1e3)
                sec[cls] = {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / PEAK_FP16_TFLOPS, 4), "ms_per_step": round(ms_c, 4), "launches_per_step": n_c,
                            "replaces": "14 launches per 64x64-level attention block: head kernel = GroupNorm-apply, conv_in, LN, "
                                        "q/k projection, V^T projection; tail kernel = out_proj+res, LN, q_proj, cross-attention, "
                                        "out_proj+res, LN, GEGLU-1, GEGLU-2+res, conv_out+res"}
            elif cls == "flash_attention":
                tf = sum(4.0 * r[1] * r[2] * r[3] * r[4] for r in rs) / P / 1e12 / (ms_c / 1e3)
                sec[cls] = {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                            "frac": round(tf / PEAK_FP16_TFLOPS, 4), "ms_per_step": round(ms_c, 4),
                            # optimistic softmax pass + exact repeat on fp16 overflow: repeats inside the timed region
                            "exact_pass_workgroups_in_timed_region": attn_exact_wg}
            else:
                gbs = sum(2.0 * r[1] * r[2] * 2 for r in rs) / P / 1e9 / (ms_c / 1e3)
                sec[cls] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(gbs / 8000.0, 4), "ms_per_step": round(ms_c, 4)}
        roofline["secondary_kernels"] = sec
        # What the matrix pipe of THIS board sustains (register-resident fp16 MFMA loop, ~20 ms, after the timed region): the
        # nominal 2.5 PF assumes the boost clock, under MFMA load the power limit sets the clock.  `frac` above stays priced
        # against the nominal peak; this is context for reading it.
        try:
            import ctypes as _C
            _tf, _ghz = _C.c_float(), _C.c_float()
            if _tsd_lib().tsd_debug_mfma_sustained(ctx.h, 20.0, _C.byref(_tf), _C.byref(_ghz)) == 0:
                roofline["mfma_sustained_probe"] = {
                    "tflops": round(_tf.value, 1), "shader_clock_ghz": round(_ghz.value, 3), "unit": "TFLOP/s",
                    "frac_of_sustained": round(achieved / _tf.value, 4) if _tf.value > 0 else None,
                    "what": "dense v_mfma_f32_16x16x32_f16 loop from registers, 4 waves/SIMD, no LDS or memory traffic"}
        except Exception as e:  # a probe must never cost the bench line
            roofline["mfma_sustained_probe"] = {"error": str(e)}
        if args.kloop:
            try:
                roofline["k_loop_model"] = k_loop_model(tsd, dev_index, in_step=[(r[0], r[1], r[2], r[3], r[4], max(0.0, r[5] - ev_overhead_ms)) for r in recs])
            except Exception as e:  # a probe must never cost the bench line
                roofline["k_loop_model"] = {"error": str(e)}
        # ---- VAE decode time (images/s end-to-end = B / (50 * step + decode)) ----
        dec_ms = None
        if dec is not None:
            sess.decode()
            ctx.synchronize()
            ctx.timer_start()
            sess.decode()
            dec_ms = ctx.timer_stop()
        ms_per_step = 1e3 * dt / K
        steps_per_s = world * K / dt
        # ---- classifier-free guidance: conditional + unconditional pass batched (2B samples per UNet call) ----
        cfg_ms = None
        if args.cfg:
            un = np.stack([tsd.rng.normal(SEED, 4000 + i, T * 768).reshape(T, 768) for i in range(lo, hi)])
            s2 = tsd.Session(unet.model, None, B, L, T, cfg=True)
            s2.set_schedule(1000, n_sched, 0)
            s2.upload(lat, cx, un, noise, 7.5)
            for i in range(2):
                s2.step(i)
            ctx.synchronize()
            ctx.timer_start()
            for i in range(K):
                s2.step((2 + i) % n_sched)
            cfg_ms = ctx.timer_stop() / K
            s2.close()
        # ---- the headline loop with peaked attention logits (the flash kernel's exact-repeat path) ----
        peaked = None
        if args.peaked:
            try:
                peaked = peaked_logits_bench(tsd, ctx, B, L, T, lat, cx, noise, n_sched, K, world * K / dt / world)
            except Exception as e:  # noqa: BLE001
                peaked = {"error": str(e)}
        # ---- BASELINE configs[4]: full-size (SD-1.5-sized) UNet, batch 4, same 50-step schedule ----
        sd15 = None
        if args.sd15:
            B5 = 4
            big = tsd.Diffusion(seed=SEED, ctx=ctx, variant="diffusion_sd15")
            s5 = tsd.Session(big.model, None, B5, L, T, cfg=False)
            s5.set_schedule(1000, n_sched, 0)
            s5.upload(lat[:B5], cx[:B5], None, noise[:, :B5])
            for i in range(3):
                s5.step(i)
            ctx.synchronize()
            ctx.timer_start()
            for i in range(K):
                s5.step((3 + i) % n_sched)
            ms5 = ctx.timer_stop() / K
            gf5 = tsd.flop_count("diffusion_sd15", L, T)
            sd15 = {"workload": f"full-size UNet (859 M parameters, 12 encoders / bottleneck / 12 decoders), latent 4x{L}x{L}, "
                                f"batch {B5}, random-init weights (BASELINE configs[4]; graph not defined by the reference)",
                    "steps_per_s": round(1e3 / ms5, 3), "ms_per_step": round(ms5, 4),
                    "algorithmic_gflop_per_step": round(gf5 * B5, 1),
                    "frac_of_fp16_mfma_peak_whole_step": round(gf5 * B5 / ms5 / PEAK_FP16_TFLOPS, 4),
                    "output_finite": bool(np.isfinite(s5.latents()).all())}
            s5.close()
            big.model.close()
        # ---- BASELINE configs[3] (img2img): VAE encoder on B x (3,512,512) + strength 0.6 of the schedule + decoder ----
        enc_ms = enc_dev_ms = None
        if args.img2img and dec is not None:
            enc = tsd.Encoder(seed=SEED, ctx=ctx)
            img = tsd.rng.uniform(SEED, 7, B * 3 * 64 * L * L, 1.0).reshape(B, 3, 8 * L, 8 * L)
            nz = tsd.rng.normal(SEED, 8, B * 4 * L * L).reshape(B, 4, L, L)
            enc.forward(img, nz)
            t0 = time.time()
            enc.forward(img, nz)  # boundary call: host buffers in and out (PCIe-inclusive)
            enc_ms = 1e3 * (time.time() - t0)
            ctx.profile_begin()   # device time of the same call: sum of its kernel launches (hipEvent pairs)
            enc.forward(img, nz)
            enc_recs = ctx.profile_records()
            ctx.profile_end()
            enc_dev_ms = sum(r[5] for r in enc_recs)
            enc.model.close()
        images_per_s = (world * B) / ((n_sched * ms_per_step + (dec_ms or 0.0)) / 1e3)
        cpu = None
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is a rank-0, N=1 figure
            t_sample, c_threads, t_numpy, c_reps = cpu_baseline(L, T)
            if t_sample is None:  # the C restatement could not be built on this host: the numpy statement is the baseline
                cpu = {"value": round(1.0 / (B * t_numpy), 6), "unit": "steps/s", "cores": c_threads, "kind": "port", "host_cores": os.cpu_count(),
                       "sample": f"one sample-step (1/{B} of a batch-{B} step, L={L}, T={T}) on the numpy oracle (im2col + BLAS, "
                                 f"{c_threads} threads) took {t_numpy:.1f} s; oracle/cref (C + OpenMP) could not be built on this host"}
            else:
                cpu = {"value": round(1.0 / (B * t_sample), 6), "unit": "steps/s", "cores": c_threads, "kind": "port",
                       "host_cores": os.cpu_count(),  # cores = the CPU quota of this process (cgroup cpu.max), the OpenMP team's size
                       "sample": f"one sample-step (1/{B} of a batch-{B} step: Diffusion.forward + DDPM update, L={L}, T={T}) "
                                 f"of oracle/cref/cref.c (the reference's loop nests in C, OpenMP over output channels / rows, "
                                 f"fp32, {c_threads} threads) took {t_sample:.2f} s (mean of {c_reps}); value = 1/({B} x that)",
                       "numpy_oracle": {"value": round(1.0 / (B * t_numpy), 6), "cores": c_threads,
                                        "sample": f"the same sample-step on the numpy oracle (im2col + BLAS) took {t_numpy:.1f} s"}}
        whole_frac = steps_per_s / world * B * total_gf / 1e3 / PEAK_FP16_TFLOPS
        assert world == args.gpus and rccl_ranks["torch_distributed"] == world, (world, args.gpus, rccl_ranks)
        line = {
            "metric": "UNet denoising steps/sec @ 512x512 latent, batch=8", "value": round(steps_per_s, 3),
            "unit": "steps/s", "n_gpus": world, "rccl_ranks": rccl_ranks, "steps": K, "warmup": W, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"Tiny-SD UNet 512x512 (latent 4x{L}x{L}), batch {B}/GPU, {n_sched}-step DDPM schedule, "
                                   f"{T}-token context, no CFG, random-init weights (BASELINE configs[1])",
                       "global_batch": world * B, "parallelism": f"dp{world} (independent prompts, weights broadcast once)"},
            "images_per_s_end_to_end": round(images_per_s, 4), "decode_ms": None if dec_ms is None else round(dec_ms, 3),
            "cfg_ms_per_step": None if cfg_ms is None else round(cfg_ms, 4),
            "img2img_config4": None if enc_ms is None else {
                "workload": f"VAE encoder on {B} x (3,{8 * L},{8 * L}) + {int(n_sched * 0.6)} UNet steps + VAE decoder (BASELINE configs[3])",
                "encode_ms_host_boundary": round(enc_ms, 3), "encode_ms_device": round(enc_dev_ms, 3), "steps": int(n_sched * 0.6),
                # images_per_s keeps its round-1 definition (encoder timed through the host boundary: upload + kernels + gaps);
                # the device-only figure has its own key
                "images_per_s": round(world * B / ((enc_ms + int(n_sched * 0.6) * ms_per_step + dec_ms) / 1e3), 4),
                "images_per_s_device_encode": round(world * B / ((enc_dev_ms + int(n_sched * 0.6) * ms_per_step + dec_ms) / 1e3), 4)},
            # the VAE halves of the hot path against the fp16 MFMA peak (algorithmic GFLOP per image: SURVEY.md Appendix B)
            "vae_roofline": {
                "decoder": None if dec_ms is None else {
                    "bound": "mfma", "ms": round(dec_ms, 3), "algorithmic_gflop": round(tsd.flop_count("decoder", L) * B, 1),
                    "achieved": round(tsd.flop_count("decoder", L) * B / dec_ms, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tsd.flop_count("decoder", L) * B / dec_ms / PEAK_FP16_TFLOPS, 4)},
                "encoder": None if enc_dev_ms is None else {
                    "bound": "mfma", "ms": round(enc_dev_ms, 3), "algorithmic_gflop": round(tsd.flop_count("encoder", 8 * L) * B, 1),
                    "achieved": round(tsd.flop_count("encoder", 8 * L) * B / enc_dev_ms, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tsd.flop_count("encoder", 8 * L) * B / enc_dev_ms / PEAK_FP16_TFLOPS, 4)}},
            "sd15_config5": sd15, "headline_peaked_logits": peaked,
            "event_ms_per_step": round(ev_ms / K, 4), "output_finite": finite,
            "frac_of_fp16_mfma_peak_whole_step": round(whole_frac, 4),
            "per_rank_steps_per_s": {"min": round(rank_rate[0], 3), "max": round(rank_rate[1], 3)},
            "weight_broadcast_s": round(bcast_s, 4), "derived_buffers_s": round(derived_s, 4), "weight_broadcast_bytes": bcast_bytes, "weight_broadcast": bcast_how,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    barrier()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
