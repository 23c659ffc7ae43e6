#!/bin/bash
# warm vs cold (rotated) weights on the M = 2048 level shapes, per ring depth (cfg 20 needs -DTSD_GEMM_EXPERIMENTAL)
for shape in "1,16,1280,1280" "0,16,1280,1280" "0,16,5120,1280"; do
  for cfg in 7 6 20 5; do
    for rot in 1 20; do
      echo -n "wrot=$rot "; TSD_BENCH_WROT=$rot TSD_BENCH_EPI=1 SHAPE=$shape,$cfg python scripts/bench_gemm1.py 2>&1 | tail -1
    done
  done
done
