#!/bin/bash
# determinism soak of the final build: every generate() result hashed (scripts/diag_race3.py), several batch sizes; then scripts/soak.py
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
out=gpurun_out/r04_soak.log; : > $out
for b in 8 1 3 5; do MODE=txt2img B=$b N=30 timeout 900 python scripts/diag_race3.py 2>&1 | tail -1 | sed "s/^/B=$b /" >> $out; done
MODE=img2img B=8 N=20 timeout 900 python scripts/diag_race3.py 2>&1 | tail -1 >> $out
STEPS=3000 timeout 900 python scripts/soak.py 2>&1 | tail -3 >> $out
B=16 STEPS=1000 timeout 900 python scripts/soak.py 2>&1 | tail -3 >> $out
N=200 timeout 600 python scripts/diag_race.py 2>&1 | tail -2 >> $out
cat $out
