#!/bin/bash
# full-size UNet (batch 4) step time under an environment switch: VAR=VAL vs unset
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-decode --sd15 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['sd15_config5']['steps_per_s'], d['sd15_config5']['ms_per_step'])"; }
for rep in 1 2; do
  echo "== off $rep"; run
  for kv in "$@"; do echo "== $kv $rep"; env $kv bash -c "$(declare -f run); run"; done
done
