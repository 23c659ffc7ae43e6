#!/bin/bash
# in-step tile sweep: SHAPE="M,N,K" GREP="M=  8192 N=  640 K=  2560" CFGS="5 45 54" [WHAT=dec] bash scripts/instep_sweep.sh  (TSD_GEMM_CFG_OVERRIDE inside a real UNet step / decode)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in $CFGS; do
  echo -n "$SHAPE cfg $c: "; TSD_GEMM_CFG_OVERRIDE="$SHAPE:$c" TOP=80 python scripts/profile_step.py | grep -E "$GREP" | awk '{print $(NF-4), $(NF-3), $(NF-1), $NF}'
done
