#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_geglu1_cfgs.txt; : > $O
for ov in "" "8192,5120,640:51" "8192,5120,640:11" "8192,5120,640:5" "8192,5120,640:54" "2048,10240,1280:51" "2048,10240,1280:11" "2048,10240,1280:5" "2048,10240,1280:54" "2048,10240,1280:45" "8192,1920,640:51" "8192,1920,640:54" "2048,3840,1280:54" "2048,3840,1280:51" ""; do
  echo "== override '$ov'" >> $O
  TSD_GEMM_CFG_OVERRIDE="$ov" WHAT=unet TOP=70 python scripts/profile_step.py 2>&1 | grep -E "total|N= 5120 K=   640|N=10240 K=  1280|N= 1920 K=   640|N= 3840 K=  1280" >> $O
done
cat $O
