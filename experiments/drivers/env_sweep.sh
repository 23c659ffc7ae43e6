#!/bin/bash
# steps/s under a list of runtime environment switches (each "VAR=VAL"), alternating with the default
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() { timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
echo -n "default: "; run
for kv in "$@"; do echo -n "$kv: "; env $kv bash -c "$(declare -f run); run"; echo -n "default: "; run; done
