#!/bin/bash
# same-kernel loop vs alternating two tile configurations on one problem (cold instruction cache probe)
for shape in "0,32,640,640" "0,16,1280,1280" "0,64,320,320"; do
  for pair in "1 3" "7 10" "0 2"; do
    set -- $pair
    a=$(TSD_BENCH_EPI=1 SHAPE=$shape,$1 python scripts/bench_gemm1.py 2>&1 | tail -1 | awk '{print $6}')
    b=$(TSD_BENCH_EPI=1 SHAPE=$shape,$2 python scripts/bench_gemm1.py 2>&1 | tail -1 | awk '{print $6}')
    ab=$(TSD_BENCH_EPI=1 TSD_BENCH_ALTCFG=$2 SHAPE=$shape,$1 python scripts/bench_gemm1.py 2>&1 | tail -1 | awk '{print $6}')
    echo "shape $shape cfg $1: $a us, cfg $2: $b us, alternating: $ab us per launch"
  done
done
