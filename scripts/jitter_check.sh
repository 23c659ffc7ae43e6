#!/bin/bash
# "Sanitizer" pass: the whole GPU suite and the determinism loops on the -DTSD_JITTER build (random wave-level delays at every tile step,
# barrier, split-K hand-off and epilogue of the GEMM, attention and fused kernels).  Correctly ordered kernels keep their bits.
#   build:  make -C stable-diffusion.mojo_amd/csrc jitter   (-> lib/libtsd_jitter.so; __graft_entry__.build() does it)
# The one-test version of this runs in the GPU suite itself: tests/test_gpu_models.py::test_jitter_build_reproduces_the_shipped_bits.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
export TSD_LIB=$PWD/stable-diffusion.mojo_amd/lib/libtsd_jitter.so
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --deselect tests/test_gpu_models.py::test_jitter_build_reproduces_the_shipped_bits 2>&1 | grep -E "passed|failed|FAILED" | tail -n 6
N=${N:-300} timeout 900 python scripts/diag_race5.py 2>&1 | tail -n 1
N=60 timeout 1200 python scripts/diag_race6.py cfg,b3,decode,encode,sd15,clip 2>&1 | tail -n 6
