#!/bin/bash
# Round-2 evidence bundle for the current build, written under gpurun_out/r02f/ (copy the summaries into profiles/):
#   bench.json                      python bench.py (all extras)                    -> the JSON line
#   kernel_stats_unet.csv           rocprofv3 --kernel-trace --stats of bench.py --no-extras (UNet step only)
#   kernel_stats_{dec,enc,sd15}.csv rocprofv3 --kernel-trace --stats of scripts/profile_step.py (decoder / encoder / full-size UNet)
#   pmc_hbm_traffic.txt             FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes (scripts/gpu_pmc_bench.sh)
#   pmc_sq.txt                      SQ counters per kernel, own --pmc pass (scripts/gpu_pmc_sq.sh)
#   shapes_{unet,dec,enc,sd15}.txt  per-launch-shape timing tables (hipEvent pairs, scripts/profile_step.py)
#   pytest_gpu.log                  python -m pytest tests -m gpu
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r02f; mkdir -p $O
cd $R; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
timeout 1500 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log
timeout 900 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
for w in unet dec enc; do WHAT=$w TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_$w.txt 2>&1; done
B=4 VARIANT=diffusion_sd15 TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_sd15.txt 2>&1
cd /tmp
rm -rf /tmp/prof_unet
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-extras > $O/prof_unet.log 2>&1
find /tmp/prof_unet -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_unet.csv \;
for w in dec enc; do
  rm -rf /tmp/prof_$w
  WHAT=$w timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o r -- python $R/scripts/profile_step.py > $O/prof_$w.log 2>&1
  find /tmp/prof_$w -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_$w.csv \;
done
rm -rf /tmp/prof_sd15
B=4 VARIANT=diffusion_sd15 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sd15 -o r -- python $R/scripts/profile_step.py > $O/prof_sd15.log 2>&1
find /tmp/prof_sd15 -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_sd15.csv \;
cd $R
bash scripts/gpu_pmc_bench.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic.txt $O/pmc_hbm_traffic.txt
bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt
head -12 $O/kernel_stats_unet.csv | cut -c1-160
head -8 $O/pmc_hbm_traffic.txt | cut -c1-150
tail -c 400 $O/bench.json
