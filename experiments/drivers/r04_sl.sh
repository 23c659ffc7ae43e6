#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cp scripts/libtsd_new.so stable-diffusion.mojo_amd/lib/libtsd.so
timeout 1200 python -m pytest tests/test_gpu_ops.py tests/test_golden.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -n 4
timeout 1500 python -m pytest tests/test_gpu_models.py -m gpu -q -p no:cacheprovider -x -k "oracle or headline or batch or session or config1" 2>&1 | tail -n 4
L=stable-diffusion.mojo_amd/lib/libtsd.so
for rep in 1 2 3; do
  for v in base new; do
    cp scripts/libtsd_$v.so $L
    echo "== $v bench $rep"; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step']['small_linear'], d['roofline']['per_class_ms_per_step']['attn_tail_chain'])"
  done
done
cp scripts/libtsd_new.so $L
