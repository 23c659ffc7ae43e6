#!/bin/bash
# round 6: the 8-wave flash attention kernel in the step - same box, interleaved: TSD_ATTN_WG8=0 (flash_attn_kernel<40,2>) vs 1 (flash_attn8_kernel)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_attn8_e2e.txt; : > $O
for r in 1 2 3 4; do for w in 0 1; do
  TSD_ATTN_WG8=$w python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['per_class_ms_per_step']
print('wg8=$w', 'steps/s', d['value'], 'ms', d['ms_per_step'], 'flash', c['flash_attention'], 'gemm', c['gemm'], 'conv', c['conv3x3'], 'chain', c['attn_tail_chain'])" >> $O
done; done
cat $O
