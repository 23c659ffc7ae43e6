#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests (one process per file so a fault cannot hide the rest),
# smoke, and a short bench.  Logs go to gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m1 gfx > gpurun_out/gpu.txt
for f in test_gpu_ops test_golden test_gpu_models; do
  timeout 1500 python -m pytest tests/$f.py -m gpu -q -p no:cacheprovider -s 2>&1 | tail -n 150 > gpurun_out/$f.log
  echo "$f exit ${PIPESTATUS[0]}" >> gpurun_out/summary.txt
done
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/summary.txt
timeout 900 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 ${BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/summary.txt
cat gpurun_out/summary.txt
grep -hE "passed|failed|error" gpurun_out/test_*.log | tail -5
tail -n 3 gpurun_out/bench.log
if [ "${PROFILE:-0}" = "1" ]; then
  cd /tmp && export TMPDIR=/tmp
  R="${GRAFT_REPO_ROOT:-/root/repo}"
  rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode > $R/gpurun_out/prof_bench.log 2>&1
  find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/kernel_stats.csv \;
  find /tmp/prof -name "*kernel_trace*.csv" -exec sh -c 'head -c 3000000 "$1" > '$R'/gpurun_out/kernel_trace_head.csv' _ {} \;
  ls -R /tmp/prof | head -20 > $R/gpurun_out/prof_ls.txt
  head -n 40 $R/gpurun_out/kernel_stats.csv
fi
