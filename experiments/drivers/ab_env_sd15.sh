#!/bin/bash
# A/B of one environment switch on one box, headline + full-size UNet (config 5): VAR=... VAL=... bash scripts/ab_env_sd15.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { timeout 900 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d.get('sd15_config5') or {}; print(d['value'], d['ms_per_step'], 'sd15', s.get('steps_per_s'), 'cfg', d.get('cfg_ms_per_step'))"; }
for rep in 1 2 3; do
  echo "== off $rep"; run
  echo "== on $rep"; env $VAR=$VAL bash -c "$(declare -f run); run"
done
