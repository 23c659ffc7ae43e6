#!/bin/bash
# round 6: GEGLU-2 + conv_out as one GEMM (TSD_FOLD_OUT) - parity of the model-level tests, then the headline A/B (same box, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_fold_ab.txt; : > $O
timeout 900 python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "oracle or composition or batch or cfg" 2>&1 | tail -n 6 >> $O
for r in 1 2 3 4; do for w in 0 1; do
  TSD_FOLD_OUT=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['per_class_ms_per_step']; l=d['roofline']['per_class_launches_per_step']
print('fold=$w', 'steps/s', d['value'], 'ms', d['ms_per_step'], 'gemm', c['gemm'], l['gemm'], 'conv', c['conv3x3'], 'flash', c['flash_attention'], 'chain', c['attn_tail_chain'])" >> $O
done; done
cat $O
