#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/keep.so
for v in ${VARIANTS:-keep}; do
  [ $v = keep ] && cp /tmp/keep.so $L || cp scripts/libtsd_$v.so $L
  for rep in 1 2 3 4; do echo "== $v txt2img ($rep)"; MODE=txt2img N=10 timeout 900 python scripts/diag_race3.py 2>&1 | tail -1 | cut -c1-250; done
done
cp /tmp/keep.so $L
