#!/bin/bash
# A/B of one environment switch on one box: VAR=value vs unset, alternating.  usage: VAR=TSD_ATTN_QB VAL=1 bash scripts/ab_env.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
if [ -n "${TESTS:-}" ]; then env $VAR=$VAL timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "$TESTS" 2>&1 | tail -n 5; fi
if [ -n "${MICRO:-}" ]; then echo "== off micro"; ITERS=50 timeout 300 python $MICRO 2>&1 | tail -n 8; echo "== on micro"; env $VAR=$VAL ITERS=50 timeout 300 python $MICRO 2>&1 | tail -n 8; fi
for rep in 1 2 3; do
  echo "== off $rep"; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  echo "== on $rep"; env $VAR=$VAL timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
done
