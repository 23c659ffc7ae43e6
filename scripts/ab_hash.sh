#!/bin/bash
# A/B on one box: two builds of libtsd.so (scripts/libtsd_base.so, scripts/libtsd_new.so): output hash of a headline-size forward
# (bitwise equality of the builds) and interleaved bench runs.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
L=stable-diffusion.mojo_amd/lib/libtsd.so
for v in base new; do cp scripts/libtsd_$v.so $L; echo "== $v"; timeout 600 python scripts/fwd_hash.py 2>&1 | tail -n 1; done
for rep in 1 2 3; do
  for v in base new; do
    cp scripts/libtsd_$v.so $L
    echo "== $v bench $rep"; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step']['attn_tail_chain'])"
  done
done
cp scripts/libtsd_new.so $L
