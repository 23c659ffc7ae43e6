#!/bin/bash
# A/B on one box: HEAD build (scripts/libtsd_base.so) against the working tree (scripts/libtsd_new.so) with the residual blocks' 1x1 skip
# convolution fused into conv2 (default) and not (TSD_RES_FUSE_SKIP=0)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
L=stable-diffusion.mojo_amd/lib/libtsd.so
cp scripts/libtsd_new.so $L
python -m pytest tests -q -m gpu -x -k "block or conv or res or model_tile" 2>&1 | grep -E "passed|failed|Error" | tail -5
b() { timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); c=d['roofline']['per_class_ms_per_step']; print(d['value'], d['ms_per_step'], 'gemm', c['gemm'], 'conv', c['conv3x3'])"; }
for rep in 1 2 3; do
  cp scripts/libtsd_base.so $L; echo -n "base      : "; b
  cp scripts/libtsd_new.so $L;  echo -n "new fuse=1: "; b
  echo -n "new fuse=0: "; TSD_RES_FUSE_SKIP=0 b
done
cp scripts/libtsd_new.so $L
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('decode_ms', d.get('decode_ms_per_batch'), 'images/s', d.get('images_per_s'))"
