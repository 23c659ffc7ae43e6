#!/bin/bash
# round 4: GroupNorm finalize with all slab partials in flight; finalize-in-apply threshold A/B; native RCCL 2 ranks on one GPU (error text)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 3 > gpurun_out/r04_gn_tests.log
VAR=TSD_GN_FINALIZE_MIN VAL=8192 bash scripts/ab_env.sh > gpurun_out/r04_gn_finalize_ab.txt 2>&1
rm -rf /tmp/profg
timeout 900 rocprofv3 --kernel-trace -d /tmp/profg -o t --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-decode --no-extras > /tmp/profg.log 2>&1
python scripts/trace_gaps.py $(find /tmp/profg -name "*kernel_trace.csv" | head -1) 160 > gpurun_out/r04_gn_trace.txt 2>&1
export TSD_BENCH_DEVICE=0 TSD_BENCH_BACKEND=gloo TSD_BENCH_NATIVE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/r04_native_2rank.txt 2>&1
echo "exit $?" >> gpurun_out/r04_native_2rank.txt
cat gpurun_out/r04_gn_tests.log gpurun_out/r04_gn_finalize_ab.txt; grep -E "k_gn_" gpurun_out/r04_gn_trace.txt | grep -v "^  +" | cut -c1-130; grep -i -E "error|duplicate|invalid|rccl|nccl" gpurun_out/r04_native_2rank.txt | head -20 | cut -c1-300
