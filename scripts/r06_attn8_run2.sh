#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/r06_attn8_run2.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "attention" 2>&1 | tail -n 8 >> $O
echo "== shipped build" >> $O
ALL=1 timeout 300 python scripts/attn8_probe.py >> $O 2>&1
echo "== ts build" >> $O
TSD_LIB=$PWD/scripts/libtsd_ts.so timeout 300 python scripts/attn8_probe.py >> $O 2>&1
for v in $TSVARS; do
  echo "== ts build, variant $v" >> $O
  TSD_LIB=$PWD/scripts/libtsd_ts$v.so ROUNDS=2 timeout 300 python scripts/attn8_probe.py >> $O 2>&1
done
for v in ${VARS:-1 2 3 16 18 32 34 48 4 8}; do
  echo "== variant $v" >> $O
  TSD_ATTN8_VAR=$v TSD_LIB=$PWD/scripts/libtsd_var.so ROUNDS=3 timeout 300 python scripts/attn8_probe.py >> $O 2>&1
done
cat $O
