#!/bin/bash
# phase timestamps for the mid-size dense GEMMs (library built with -DTSD_GEMM_TS at scripts/libtsd_ts.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp scripts/libtsd_ts.so stable-diffusion.mojo_amd/lib/libtsd.so
for shape in ${SHAPES:-"0,64,320,320,-1" "0,32,640,640,-1" "0,16,1280,1280,-1" "0,64,320,2560,-1"}; do
  for e in ${EPIS:-0 1}; do
  echo "== shape $shape epi=$e"; TSD_GEMM_TS=1 TSD_BENCH_EPI=$e SHAPE=$shape timeout 120 python scripts/bench_gemm1.py 2>&1 | grep -E "\[ts\]|TF"
  done
done
