#!/bin/bash
# Look for a box on which the 10-generate stress test fails in fresh processes; there, localise: which step diverges first (diag_race4.py),
# and which run-time switch makes it go away.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
one() { env "$@" timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "bitwise_repeatable_under_stress" 2>&1 | tail -n 1 | grep -oE "(passed|failed)" | head -1; }
f=0; for i in 1 2 3 4 5 6; do r=$(one X=1); [ "$r" = failed ] && f=$((f+1)); done
echo "probe: $f of 6 failed"
if [ $f -ge 2 ]; then
  for i in 1 2 3 4; do EVERY=5 N=10 timeout 600 python scripts/diag_race4.py 2>&1 | tail -n 6; done
  for v in X=1 TSD_CHAIN=0 TSD_GEMM_SPLITK=0 TSD_ATTN_QB=1 TSD_GEMM_TUNE=0 TSD_QKV_FUSE=0 TSD_GN_COMPOSITE=0 TSD_RES_FUSE_SKIP=0; do
    for i in 1 2 3 4 5 6 7 8; do echo "$v $(one $v)"; done
  done | sort | uniq -c
fi
