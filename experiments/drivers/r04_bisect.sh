#!/bin/bash
# usage: LIBS="A B C" bash scripts/r04_bisect.sh : 30 txt2img generate() calls per library, count the distinct results
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
L=stable-diffusion.mojo_amd/lib/libtsd.so
cp $L /tmp/libtsd_keep.so
for rep in 1 2; do for v in $LIBS; do
  cp scripts/libtsd_$v.so $L
  echo -n "== $v: "; MODE=txt2img N=${N:-20} timeout 600 python scripts/diag_race3.py 2>&1 | tail -n 1
done; done
cp /tmp/libtsd_keep.so $L
