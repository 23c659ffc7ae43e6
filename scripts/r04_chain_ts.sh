#!/bin/bash
# phase timestamps of the fused tail / head kernels (scripts/libtsd_ts.so = a -DTSD_CHAIN_TS build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/keep.so; cp scripts/libtsd_ts.so $L
KIND=0 timeout 300 python scripts/chain_ts.py > gpurun_out/r04_chain_ts_tail.txt 2>&1
KIND=1 timeout 300 python scripts/chain_ts.py > gpurun_out/r04_chain_ts_head.txt 2>&1
cp /tmp/keep.so $L
cat gpurun_out/r04_chain_ts_tail.txt gpurun_out/r04_chain_ts_head.txt
