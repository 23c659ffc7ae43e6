#!/bin/bash
# Decompose GEMM/conv time: full vs DMA-only vs compute-only (TSD_GEMM_DBG experiment knob).
for shape in "1,64,320,320" "1,64,640,320" "1,32,640,640" "1,16,1280,1280" "0,64,320,320" "0,64,1280,320" "0,64,320,2560"; do
  for cfg in ${CFGS:-0 5 12 11}; do
    for dbg in 0 1 2 3; do
      echo -n "dbg=$dbg "; TSD_GEMM_DBG=$dbg SHAPE=$shape,$cfg python scripts/bench_gemm1.py 2>&1 | tail -1
    done
  done
done
