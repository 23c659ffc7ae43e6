"""Diagnosis: the library's own RCCL path (tsd_dist_*, librccl via dlopen) in a process where torch's HIP runtime / RCCL is
already loaded and initialised.  Prints which librccl objects are mapped and whether ncclCommInitRank succeeds."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-diffusion.mojo_amd")); sys.path.insert(0, ROOT)
mode = os.environ.get("MODE", "torch_first")
if mode == "torch_first":
    import torch
    torch.zeros(4, device="cuda:0").sum().item()
    if os.environ.get("TORCH_DIST") == "1":
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", RANK="0", WORLD_SIZE="1")
        dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
        t = torch.ones(8, device="cuda:0"); dist.broadcast(t, 0); torch.cuda.synchronize()
import tsd
from tsd._lib import check, lib
tsd.set_strict(True)
enc = tsd.Encoder(seed=1)
uid = C.create_string_buffer(128)
rc = lib().tsd_dist_unique_id(uid)
print("unique_id rc", rc, lib().tsd_last_error() if rc else "")
rc = lib().tsd_dist_init(enc.model.ctx.h, 0, 1, uid)
print("dist_init rc", rc, (lib().tsd_last_error() or b"").decode() if rc else "")
if rc == 0:
    rc = lib().tsd_dist_broadcast_weights(enc.model.h, 0)
    print("broadcast rc", rc)
maps = sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "amdhip64" in l})
print("mapped:", maps)
