#!/bin/bash
# upper bound of what fusing norm launches away could buy: TSD_EXP_SKIP bits (1: GN2 of residual blocks, 2: mid-level LayerNorms, 4: GN1) skip the launches (wrong numbers, timing only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env TSD_EXP_SKIP=$1 timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('skip=$1', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do for s in 0 1 2 4 7; do run $s; done; done
