#!/bin/bash
# same-box A/B of two library builds on the full-size UNet (batch 4) and the headline: scripts/r05_ab_sd15.sh <libA.so> <libB.so> [rounds]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do for L in $A $B; do
  TSD_LIB=$PWD/$L python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-cfg --no-img2img --no-peaked --no-kloop --no-decode 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L'.split('/')[-1], 'headline', d['value'], 'sd15 steps/s', d['sd15_config5']['steps_per_s'])"
done; done
