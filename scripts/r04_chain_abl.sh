#!/bin/bash
# ablations of the fused tail's GEGLU pipeline (timing only): FFN phase ticks per build (scripts/libtsd_abl<bits>.so = -DTSD_CHAIN_TS -DTSD_CHAIN_ABL=<bits>)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so; cp $L /tmp/keep.so
out=gpurun_out/r04_chain_ablation.txt; : > $out
for v in ts abl1 abl2 abl4 abl8 abl16 abl33 abl6 abl14; do
  [ -f scripts/libtsd_$v.so ] || continue
  cp scripts/libtsd_$v.so $L
  echo "== $v" >> $out
  KIND=0 timeout 300 python scripts/chain_ts.py 2>&1 | grep -A11 "round 2" | grep -E "FFN \(|total|S1 gemm" >> $out
done
cp /tmp/keep.so $L; cat $out
