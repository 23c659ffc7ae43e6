#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_fold_cfgs.txt; : > $O
WHAT=unet TOP=30 python scripts/profile_step.py >> $O 2>&1
for ov in "" "8192,640,3200:5" "8192,640,3200:51" "8192,640,3200:11" "8192,640,3200:0" "2048,1280,6400:47" "2048,1280,6400:45" "2048,1280,6400:5" "2048,1280,6400:7"; do
  echo "== override '$ov'" >> $O
  TSD_GEMM_CFG_OVERRIDE="$ov" WHAT=unet TOP=60 python scripts/profile_step.py 2>&1 | grep -E "total|N=  640 K=  3200|N= 1280 K=  6400" >> $O
done
cat $O
