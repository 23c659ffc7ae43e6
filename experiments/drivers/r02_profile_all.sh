#!/bin/bash
# Round-2 baseline evidence: per-shape launch tables (UNet step, decoder, encoder, full-size UNet) and rocprofv3
# kernel-trace stats of the decoder / encoder / full-size UNet.  Output under gpurun_out/r02/.
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/${TAG:-r02}; mkdir -p $O
cd $R; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
for w in unet dec enc; do WHAT=$w TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_$w.txt 2>&1; done
B=4 VARIANT=diffusion_sd15 TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_sd15.txt 2>&1
cd /tmp
for w in dec enc; do
  rm -rf /tmp/prof_$w
  WHAT=$w timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o r -- python $R/scripts/profile_step.py > $O/prof_$w.log 2>&1
  find /tmp/prof_$w -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_$w.csv \;
done
rm -rf /tmp/prof_sd15
B=4 VARIANT=diffusion_sd15 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sd15 -o r -- python $R/scripts/profile_step.py > $O/prof_sd15.log 2>&1
find /tmp/prof_sd15 -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_sd15.csv \;
cd $R
timeout 900 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
