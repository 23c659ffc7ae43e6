#!/bin/bash
# A/B/C on one box: alternate the builds named on the command line (scripts/libtsd_<name>.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so
for rep in 1 2 3; do
  for v in "$@"; do
    cp scripts/libtsd_$v.so $L
    echo -n "$v $rep: "; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
  done
done
