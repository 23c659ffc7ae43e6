#!/bin/bash
# Does any result depend on memory nobody wrote?  The same generate() in fresh processes with fresh device allocations filled with different bytes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
for pz in -1 0 60 123 255 -1 255; do
  echo -n "poison $pz: "; TSD_DEBUG_POISON=$pz MODE=${MODE:-txt2img} N=2 timeout 600 python scripts/diag_race3.py 2>&1 | tail -n 1
done
TSD_DEBUG_POISON=255 timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 4
