#!/bin/bash
# same-box A/B: the VAE's fused-skip 256 -> 128 residual block on the 128x128 tile (TSD_GEMM_SKIP128=1, round-5 default) against the 64-row detour
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for r in 1 2 3; do for v in 0 1; do
  TSD_GEMM_SKIP128=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfg --no-sd15 --no-peaked --no-kloop 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SKIP128=$v', 'decode_ms', d['decode_ms'], 'encode_dev_ms', d['img2img_config4']['encode_ms_device'], 'steps/s', d['value'])"
done; done
