#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_flip_old.txt; : > $O
for r in 1 2 3; do for L in "" $PWD/scripts/libtsd_flip.so; do
  echo "== lib=${L:-shipped}" >> $O
  TSD_LIB=$L TSD_ATTN_WG8=0 ITERS=30 timeout 300 python scripts/bench_attn.py >> $O 2>&1
done; done
cat $O
