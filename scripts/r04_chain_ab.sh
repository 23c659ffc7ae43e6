#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cp scripts/libtsd_new.so stable-diffusion.mojo_amd/lib/libtsd.so
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_golden.py tests/test_gpu_range.py -m gpu -q -p no:cacheprovider -x -k "attn or attention or golden or geglu or heavy or two_contexts" 2>&1 | tail -n 5 > gpurun_out/r04_tail_tests.log
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x -k "fused or headline or batch or forward or stress" 2>&1 | tail -n 5 >> gpurun_out/r04_tail_tests.log
bash scripts/ab.sh > gpurun_out/r04_tail_pinned_ab2.txt 2>&1
bash scripts/r04_chain_ts.sh > /dev/null 2>&1
cat gpurun_out/r04_tail_tests.log gpurun_out/r04_tail_pinned_ab2.txt; grep -A12 "round 2" gpurun_out/r04_chain_ts_tail.txt | head -14
