#!/bin/bash
# per-block phase timestamps of the GEMM kernel (library built with -DTSD_GEMM_TS)
for e in 0 1; do
for shape in "0,64,320,320,0" "0,64,320,320,11" "0,64,320,2560,0" "0,16,1280,1280,7" "1,64,320,320,0" "1,16,1280,1280,6"; do
  echo "== shape $shape epi=$e"; TSD_GEMM_TS=1 TSD_BENCH_EPI=$e SHAPE=$shape python scripts/bench_gemm1.py 2>&1 | grep -E "\[ts\]|TF"
done; done
