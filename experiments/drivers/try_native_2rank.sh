#!/bin/bash
# does the library's own RCCL broadcast (tsd_dist_*) run with nranks = 2 when both ranks sit on this box's ONE GPU?
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
export TSD_BENCH_DEVICE=0 TSD_BENCH_BACKEND=gloo TSD_BENCH_NATIVE_DIST=1 HSA_ENABLE_IPC_MODE_LEGACY=0
for extra in "" "NCCL_IGNORE_DUPLICATE_GPU=1" ; do
  echo "== extra env: '$extra'"
  env $extra timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29561 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2>&1 | grep -v "^\s*$" | tail -n 12 | cut -c1-400
  echo "exit ${PIPESTATUS[0]}"
done
