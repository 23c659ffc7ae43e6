#!/bin/bash
# round 4, first GPU call: GPU tests (incl. the new range / two-context tests), smoke, a bench line on this box, the any-order probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
./scripts/micro/anyorder > gpurun_out/r04_anyorder.txt 2>&1
for f in test_gpu_range test_gpu_ops test_golden test_gpu_models; do
  timeout 1800 python -m pytest tests/$f.py -m gpu -q -p no:cacheprovider -s 2>&1 | tail -n 250 > gpurun_out/r04_$f.log
  echo "$f exit ${PIPESTATUS[0]}" >> gpurun_out/r04_summary.txt
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/r04_summary.txt
timeout 900 python bench.py --steps 100 --warmup 10 > gpurun_out/r04_bench_v1.log 2>&1; echo "bench exit $?" >> gpurun_out/r04_summary.txt
cat gpurun_out/r04_summary.txt gpurun_out/r04_anyorder.txt
grep -hE "passed|failed|error" gpurun_out/r04_test_*.log | tail -8
tail -n 2 gpurun_out/r04_bench_v1.log | cut -c1-600
