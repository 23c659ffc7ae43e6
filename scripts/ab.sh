#!/bin/bash
# A/B on one box: alternate two builds of libtsd.so (scripts/libtsd_base.so, scripts/libtsd_new.so).
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
L=stable-diffusion.mojo_amd/lib/libtsd.so
for rep in 1 2 3; do
  for v in base new; do
    cp scripts/libtsd_$v.so $L
    if [ "$rep" = 1 ] && [ -n "${MICRO:-}" ]; then echo "== $v micro"; ITERS=50 timeout 300 python $MICRO 2>&1 | tail -n 8; fi
    echo "== $v bench $rep"; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step']['attn_tail_chain'])"
  done
done
cp scripts/libtsd_new.so $L
