#!/bin/bash
# round 6: what are warm weights worth in the step?  TSD_WWARM=1 puts a touch kernel (same stream) in front of GEMM i that reads the weights of
# GEMM i+1; the per-class GEMM / conv time of the profiler (which brackets the GEMM launches only) is the ceiling of an in-kernel scheme
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_wwarm_probe.txt; : > $O
for r in 1 2 3; do for w in 0 1; do
  TSD_WWARM=$w python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['per_class_ms_per_step']
print('wwarm=$w', 'steps/s', d['value'], 'ms', d['ms_per_step'], 'gemm', c['gemm'], 'conv', c['conv3x3'], 'flash', c['flash_attention'], 'chain', c['attn_tail_chain'], 'gn', c['groupnorm'])" >> $O
done; done
cat $O
