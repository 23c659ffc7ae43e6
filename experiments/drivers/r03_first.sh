#!/bin/bash
# Round-3 first GPU call: parity with the tightened bounds + the new headline-size oracle tests, a bench line, the
# per-dispatch kernel trace of the step (scripts/trace_gaps.py: in-step duration and idle gap per launch), the LDS read-rate
# microbenchmark, and the back-to-back timing of the small GEMMs for the in-step comparison.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
./scripts/micro/lds_read_rate > $O/lds_read_rate.txt 2>&1; cat $O/lds_read_rate.txt
timeout 2400 python -m pytest tests -q -m gpu -s -x > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
grep "\[parity\]" $O/pytest_gpu.log > $O/parity.log
timeout 900 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
WHAT=unet TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_unet.txt 2>&1
cd /tmp; rm -rf /tmp/prof_unet
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-extras > $O/prof_unet.log 2>&1
find /tmp/prof_unet -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_unet.csv \;
find /tmp/prof_unet -name "*kernel_trace*.csv" -exec cp {} $O/kernel_trace_unet.csv \;
cd $R
python scripts/trace_gaps.py $O/kernel_trace_unet.csv 160 > $O/trace_gaps.txt 2>&1; head -40 $O/trace_gaps.txt | cut -c1-200
# back-to-back microbench of the small GEMMs: dense M N K, dispatcher's configuration, with and without weight rotation
python - <<'EOF' > $O/gemm_b2b.txt 2>&1
import os, sys, ctypes as C
sys.path.insert(0, "stable-diffusion.mojo_amd"); sys.path.insert(0, ".")
import tsd
from tsd._lib import lib
L = lib(); ctx = tsd.default_context()
def bench(conv, B, H, W, Cin, N, cfg=-1, iters=50, wrot=1, epi=0):
    os.environ["TSD_BENCH_WROT"] = str(wrot); os.environ["TSD_BENCH_EPI"] = str(epi)
    ms = C.c_float()
    r = L.tsd_debug_gemm_bench(ctx.h, conv, B, H, W, Cin, N, 1, 0, cfg, iters, C.byref(ms))
    return ms.value * 1e3 if r == 0 else float("nan")
for (M, N, K) in ((2048, 1280, 1280), (8192, 640, 640), (8192, 5120, 640), (2048, 10240, 1280), (2048, 1280, 5120), (8192, 640, 2560), (2048, 2560, 1280), (8192, 1280, 640)):
    for wrot in (1, 40):
        for epi in (0, 1):
            us = bench(0, 1, M, 1, K, N, -1, 60, wrot, epi)
            print(f"gemm {M}x{N}x{K} wrot={wrot} epi={epi}: {us:7.2f} us  {2.0*M*N*K/us/1e6:7.1f} TF")
EOF
cat $O/gemm_b2b.txt
