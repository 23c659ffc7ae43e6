"""Per-block phase timestamps of a few GEMM launches on a -DTSD_GEMM_TS build (make BUILD=build_ts OUT=../lib/libtsd_ts.so EXTRA=-DTSD_GEMM_TS):
   TSD_LIB=$PWD/stable-diffusion.mojo_amd/lib/libtsd_ts.so python scripts/gemm_ts_probe.py      (profiles/r05_gemm_ts.txt)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"), "stable-diffusion.mojo_amd"))
os.environ["TSD_BENCH_EPI"]="1"; os.environ["TSD_GEMM_TS"]="1"
import tsd
from tsd._lib import lib
ctx = tsd.Context(0)
for (conv,B,H,W,cin,N,cfg) in ((0,8,32,32,640,640,54),(0,8,32,32,640,640,5),(0,8,16,16,1280,1280,47),(0,8,32,32,2560,640,54),(1,8,64,64,320,320,51)):
    ms=C.c_float()
    print("== conv",conv,"M",B*H*W,"N",N,"Cin",cin,"cfg",cfg, flush=True)
    r=lib().tsd_debug_gemm_bench(ctx.h, conv,B,H,W,cin,N,1,0,cfg,20,C.byref(ms))
    print("   ->", r, round(ms.value*1e3,2),"us", flush=True)
