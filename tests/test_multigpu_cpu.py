"""N > 1 path on CPU: two `gloo` processes exercise bench.py's sharding and the weight-broadcast protocol
(rank 0 owns the blob, every other rank receives identical bytes, batch shards are disjoint and cover the
global batch).  No GPU needed; the RCCL transport itself is exercised by the driver's multi-GPU run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
    import bench
    import tsd.rng as prng
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # weight blob: rank 0 generates, everyone else starts from garbage and must end up bit-identical - through the REAL
    # bench.broadcast_weights (the function the multi-GPU bench runs), on a host "blob" with the gloo backend
    n = 1 << 16
    ref = prng.uniform(1234, 4096 + 7, n, 0.05).astype(np.float32)
    blob = ref.copy() if rank == 0 else np.full(n, np.nan, dtype=np.float32)

    class HostModel:  # stands in for tsd.Model: same two methods broadcast_weights uses
        loaded = False

        def packed_blob(self):
            return blob.ctypes.data, blob.nbytes

        def mark_loaded(self):
            self.loaded = True

    hm = HostModel()
    secs, nbytes = bench.broadcast_weights([hm], rank, world, "cpu", "gloo")
    ok_blob = bool(np.array_equal(blob, ref)) and nbytes == blob.nbytes and hm.loaded == (rank != 0)
    # an injected failure must raise on every rank (bench.py has no silent per-rank fallback any more)
    os.environ["TSD_BENCH_FAIL_BCAST"] = "1"
    try:
        bench.broadcast_weights([hm], rank, world, "cpu", "gloo")
        ok_blob = False
    except RuntimeError:
        pass
    del os.environ["TSD_BENCH_FAIL_BCAST"]
    # the native-RCCL path's control step: rank 0's 128-byte unique id reaches every rank (bench.exchange_unique_id);
    # make_id must run on rank 0 only
    calls = []
    uid = bench.exchange_unique_id(rank, lambda: (calls.append(1), bytes(range(128)))[1])
    ok_blob = ok_blob and uid == bytes(range(128)) and len(calls) == (1 if rank == 0 else 0)
    # batch shards of a global batch of world*8 prompts
    lo, hi = bench.shard_range(world * 8, rank, world)
    ids = torch.zeros(world * 8, dtype=torch.int64)
    ids[lo:hi] = 1
    dist.all_reduce(ids)
    # timing reduction = MAX over ranks (bench contract)
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.barrier()
    out[rank] = (ok_blob, hi - lo, bool((ids == 1).all()), float(t.item()))
    dist.destroy_process_group()


def test_two_rank_sharding_and_broadcast():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    for r in range(world):
        ok_blob, nshard, cover, tmax = out[r]
        assert ok_blob and nshard == 8 and cover and tmax == 2.0


@pytest.mark.parametrize("total,world", [(64, 8), (16, 2), (8, 1), (10, 3)])
def test_shard_range_partitions(total, world):
    sys.path.insert(0, ROOT)
    import bench
    seen = []
    for r in range(world):
        lo, hi = bench.shard_range(total, r, world)
        seen += list(range(lo, hi))
    assert seen == list(range(total))


# ---- bench.py --gpus N started plainly: it launches its own N ranks and refuses to print a line about another N --------------------
def _run_bench(args, env_extra, timeout=300):
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, timeout=timeout)


def test_bench_gpus_2_started_plainly_launches_two_ranks():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: two ranks rendezvous over gloo on 127.0.0.1 and rank 0 reports
    n_gpus = 2 (TSD_BENCH_LAUNCH_ONLY=1 stops before the first GPU call - there is no GPU here)."""
    import json
    r = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"TSD_BENCH_LAUNCH_ONLY": "1", "TSD_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == {"torch_distributed": 2, "counted": 2}
    assert line["shards"] == [[0, 8], [8, 16]]
    assert "starting 2 ranks" in r.stderr


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run_bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "TSD_BENCH_LAUNCH_ONLY": "1"}, timeout=60)
    assert r.returncode == 2 and "refusing to run" in r.stderr, (r.returncode, r.stderr[-500:])
    r = _run_bench(["--gpus", "4"], {"WORLD_SIZE": "1", "RANK": "0", "TSD_BENCH_LAUNCH_ONLY": "1"}, timeout=60)
    assert r.returncode == 2 and "refusing to run" in r.stderr, (r.returncode, r.stderr[-500:])


def test_bench_refuses_more_gpus_than_are_visible():
    """No HIP device in this container: `--gpus 2` started plainly must exit non-zero before it launches anything."""
    sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd"))
    from tsd._lib import lib
    if lib().tsd_device_count() >= 2:
        pytest.skip("two devices visible")
    r = _run_bench(["--gpus", "2"], {}, timeout=120)
    assert r.returncode == 2 and "HIP device(s) visible: refusing to run" in r.stderr, (r.returncode, r.stderr[-500:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
