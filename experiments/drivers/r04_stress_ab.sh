#!/bin/bash
# usage: LIBS="A B" REPS=3 bash scripts/r04_stress_ab.sh : the 10-generate stress test REPS times per library, round robin twice
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
L=stable-diffusion.mojo_amd/lib/libtsd.so
cp $L /tmp/libtsd_keep.so
for round in 1 2; do for v in $LIBS; do
  cp scripts/libtsd_$v.so $L
  echo -n "== $v:"
  for i in $(seq ${REPS:-3}); do timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -k "bitwise_repeatable_under_stress" 2>&1 | tail -n 1 | grep -oE "[0-9]+ (passed|failed)" | tr '\n' ' '; done; echo
done; done
cp /tmp/libtsd_keep.so $L
