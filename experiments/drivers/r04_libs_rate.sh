#!/bin/bash
# rate of run-to-run differences (scripts/diag_race5.py) for a list of library builds scripts/libtsd_<name>.so
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/libtsd_keep.so
for v in $LIBS; do cp scripts/libtsd_$v.so $L; echo -n "$v: "; N=${N:-500} timeout 1200 python scripts/diag_race5.py 2>&1 | tail -n 1 | cut -c1-160; done
cp /tmp/libtsd_keep.so $L
