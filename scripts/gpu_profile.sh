#!/bin/bash
# rocprofv3 kernel-trace stats of bench.py on the GPU box -> gpurun_out/kernel_stats.csv
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-extras > $R/gpurun_out/prof_bench.log 2>&1
find /tmp/prof -name "*kernel_stats*.csv" -exec cp {} $R/gpurun_out/kernel_stats.csv \;
head -n 30 $R/gpurun_out/kernel_stats.csv | cut -c1-150
