"""One sample-step of the C restatement at several OpenMP team sizes on this host (which team size does the box allow?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import cref, models, rng, spec
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "available", cref.threads_available())
for p in ("/sys/fs/cgroup/cpu.max",):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, e)
print(open("/proc/loadavg").read().strip())
P = spec.init_params("diffusion", 1234, only_used=True)
L = 64
lat = rng.normal(1234, 2, 4 * L * L).reshape(4, L, L); ctx = rng.normal(1234, 5, 77 * 768).reshape(77, 768)
cb = cref.backend()
for n in (int(a) for a in sys.argv[1:]):
    cref.set_threads(n)
    t0 = time.time()
    with models.using_ops(cb):
        y = models.diffusion(P, lat, ctx, cb.time_embedding(980.0))
    print(f"threads {cref.threads():4d}: {time.time() - t0:.2f} s", flush=True)
