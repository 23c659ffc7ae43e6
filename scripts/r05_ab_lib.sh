#!/bin/bash
# same-box A/B of two builds of the library: scripts/r05_ab_lib.sh <libA.so> <libB.so> [rounds]   (headline loop, 100 steps, interleaved)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do for L in $A $B; do
  TSD_LIB=$PWD/$L python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --no-decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['roofline']['per_class_ms_per_step']
print('$L'.split('/')[-1], 'steps/s', d['value'], 'flash', c['flash_attention'], 'gemm', c['gemm'], 'conv', c['conv3x3'], 'chain', c['attn_tail_chain'])"
done; done
