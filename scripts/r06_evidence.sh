#!/bin/bash
# Round-6 evidence bundle for the CURRENT build, written under gpurun_out/r06f/ (copy the summaries into profiles/):
#   pytest_gpu.log / parity.log      python -m pytest tests -m gpu -s
#   bench.json                       python bench.py (all extras, cpu_baseline)
#   kloop_probe.txt                  fixed + slope x K-tiles per tile configuration, least squares + spreads (scripts/kloop_probe.py = bench.k_loop_model)
#   attn8_probe.txt                  the 4096 x 4096 d = 40 call: 4-wave against 8-wave kernel, interleaved (scripts/attn8_probe.py)
#   shapes_{unet,dec,enc,sd15}.txt   per-launch-shape timing tables (hipEvent pairs, scripts/profile_step.py)
#   kernel_stats_{unet,dec,enc}.csv  rocprofv3 --kernel-trace --stats summaries
#   pmc_hbm_traffic.txt              FETCH_SIZE / WRITE_SIZE per kernel, separate --pmc passes (TCC slots)
#   pmc_sq.txt, pmc_tcp.txt          SQ / vL1D counters per kernel (own passes)
#   pmc_lds.txt                      LDS side: SQ_INSTS_LDS, SQ_ACTIVE_INST_LDS, SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT, FIFO-full counters
#   pmc_tcc.txt                      L2: TCC_REQ / TCC_HIT / TCC_MISS per kernel
#   pmc_clock.txt                    GRBM_GUI_ACTIVE per dispatch next to the same pass's kernel duration -> effective clock
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; O=$R/gpurun_out/r06f; mkdir -p $O
cd $R; export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
timeout 2400 python -m pytest tests -q -m gpu -s > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
grep "\[parity\]" $O/pytest_gpu.log > $O/parity.log; tail -1 $O/pytest_gpu.log >> $O/parity.log
timeout 900 python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err
python scripts/kloop_probe.py > $O/kloop_probe.txt 2>&1   # bench.py's own probe list through the same function: the two must agree within the spreads
ALL=1 ROUNDS=5 timeout 300 python scripts/attn8_probe.py > $O/attn8_probe.txt 2>&1
for w in unet dec enc; do WHAT=$w TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_$w.txt 2>&1; done
B=4 VARIANT=diffusion_sd15 TOP=80 timeout 600 python scripts/profile_step.py > $O/shapes_sd15.txt 2>&1
cd /tmp
rm -rf /tmp/prof_unet
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_unet -o r -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-decode --no-extras > $O/prof_unet.log 2>&1
find /tmp/prof_unet -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_unet.csv \;
find /tmp/prof_unet -name "*kernel_trace*.csv" -exec cp {} /tmp/kernel_trace_unet.csv \;
python $R/scripts/trace_gaps.py /tmp/kernel_trace_unet.csv > $O/trace_per_dispatch.txt 2>&1
for w in dec enc; do
  rm -rf /tmp/prof_$w
  WHAT=$w timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$w -o r -- python $R/scripts/profile_step.py > $O/prof_$w.log 2>&1
  find /tmp/prof_$w -name "*kernel_stats*.csv" -exec cp {} $O/kernel_stats_$w.csv \;
done
cd $R
bash scripts/gpu_pmc_bench.sh > /dev/null 2>&1; cp gpurun_out/pmc_traffic.txt $O/pmc_hbm_traffic.txt
bash scripts/gpu_pmc_sq.sh > /dev/null 2>&1; cp gpurun_out/pmc_sq.txt $O/pmc_sq.txt
bash scripts/gpu_pmc_tcp.sh > /dev/null 2>&1; cp gpurun_out/pmc_tcp.txt $O/pmc_tcp.txt
bash scripts/gpu_pmc_generic.sh lds "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES" > /dev/null 2>&1; cp gpurun_out/pmc_lds.txt $O/pmc_lds.txt
bash scripts/gpu_pmc_generic.sh tcc "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" > /dev/null 2>&1; cp gpurun_out/pmc_tcc.txt $O/pmc_tcc.txt
bash scripts/gpu_pmc_generic.sh clock "GRBM_GUI_ACTIVE GRBM_COUNT" > /dev/null 2>&1; cp gpurun_out/pmc_clock.txt $O/pmc_clock.txt
head -12 $O/kernel_stats_unet.csv | cut -c1-160
head -8 $O/pmc_hbm_traffic.txt | cut -c1-150
head -8 $O/pmc_tcc.txt | cut -c1-150
head -8 $O/pmc_clock.txt | cut -c1-150
tail -c 400 $O/bench.json
