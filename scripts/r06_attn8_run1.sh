#!/bin/bash
# round 6, first GPU look at the 8-wave flash attention kernel: parity, then interleaved timing of every variant build
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_attn8_run1.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -n 25 >> $O
echo "== shipped build" >> $O
ALL=1 timeout 300 python scripts/attn8_probe.py >> $O 2>&1
echo "== ts build" >> $O
TSD_LIB=$PWD/scripts/libtsd_ts.so timeout 300 python scripts/attn8_probe.py >> $O 2>&1
for v in 1 2 3 16 17 4 8 12; do
  echo "== variant $v" >> $O
  TSD_ATTN8_VAR=$v TSD_LIB=$PWD/scripts/libtsd_var.so ROUNDS=3 timeout 300 python scripts/attn8_probe.py >> $O 2>&1
done
cat $O
