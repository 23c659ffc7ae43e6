#!/bin/bash
# round 4: price a GEMM -> GEMM seam kept inside one launch (TSD_EXP_PAIR=1, timing only: no dependency wait at all)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_range.py -m gpu -q -p no:cacheprovider 2>&1 | tail -n 5 > gpurun_out/r04_range_v2.log
out=gpurun_out/r04_pair_ab.txt; : > $out
for rep in 1 2 3; do
  echo "== separate launches $rep" >> $out; timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step'])" >> $out
  echo "== pairs (timing only) $rep" >> $out; TSD_EXP_PAIR=1 timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-decode --no-extras 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline']['per_class_ms_per_step'])" >> $out
done
for mode in 0 1; do
  rm -rf /tmp/prof$mode
  TSD_EXP_PAIR=$mode timeout 900 rocprofv3 --kernel-trace -d /tmp/prof$mode -o t --output-format csv -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-decode --no-extras > /tmp/prof$mode.log 2>&1
  f=$(find /tmp/prof$mode -name "*kernel_trace.csv" | head -1)
  python scripts/trace_gaps.py $f 160 > gpurun_out/r04_pair_trace_$mode.txt 2>&1
done
cat $out; cat gpurun_out/r04_range_v2.log
