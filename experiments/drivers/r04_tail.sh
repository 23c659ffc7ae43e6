#!/bin/bash
# round 4: fused attention-block kernels on a 1 x 4 wave grid (64 x 80 wave tiles): parity, then same-box A/B against the 2 x 2 build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cp scripts/libtsd_new.so stable-diffusion.mojo_amd/lib/libtsd.so
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_golden.py tests/test_gpu_range.py -m gpu -q -p no:cacheprovider -x -k "attn or attention or golden or geglu or heavy or two_contexts" 2>&1 | tail -n 25 > gpurun_out/r04_tail_tests.log
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 25 >> gpurun_out/r04_tail_tests.log
bash scripts/ab.sh > gpurun_out/r04_tail_1x4_ab.txt 2>&1
cat gpurun_out/r04_tail_tests.log gpurun_out/r04_tail_1x4_ab.txt
