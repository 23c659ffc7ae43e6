#!/bin/bash
# "Sanitizer" pass: the whole GPU suite and the determinism loops on a -DTSD_JITTER build (random wave-level delays at every tile step,
# barrier, split-K hand-off and epilogue of the GEMM, attention and fused kernels).  Correctly ordered kernels keep their bits.
#   build:  cp -r stable-diffusion.mojo_amd/csrc /tmp/tJ/pkg/csrc (+ include), make EXTRA=-DTSD_JITTER, cp the .so to scripts/libtsd_J2.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/libtsd_keep.so; cp scripts/libtsd_J2.so $L
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -E "passed|failed|FAILED" | tail -n 6
N=300 timeout 900 python scripts/diag_race5.py 2>&1 | tail -n 1
N=60 timeout 1200 python scripts/diag_race6.py cfg,b3,decode,encode,sd15,clip 2>&1 | tail -n 6
cp /tmp/libtsd_keep.so $L
