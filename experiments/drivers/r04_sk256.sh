#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_sk256_ab.txt; : > $out
for v in 3; do TSD_GEMM_SK256=$v timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x -k "res or conv or forward_matches or headline or splitk or batch" 2>&1 | tail -n 3 >> $out; done
for rep in 1 2 3; do
  for v in 0 1 2 3; do
    echo "== TSD_GEMM_SK256=$v ($rep)" >> $out
    TSD_GEMM_SK256=$v timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step']['conv3x3'], d['roofline']['per_class_ms_per_step']['gemm'])" >> $out
  done
done
cat $out
