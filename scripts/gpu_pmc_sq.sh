#!/bin/bash
# SQ counters per kernel of the bench step (own --pmc pass, kernel-trace only): MFMA busy, VALU/LDS activity, waits,
# LDS bank conflicts.  Output: gpurun_out/pmc_sq.txt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/pmc_sq -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-extras > $R/gpurun_out/pmc_sq.log 2>&1
find /tmp/pmc_sq -name "*counter_collection*.csv" -exec cp {} /tmp/pmc_sq.csv \;
python3 - <<PY > $R/gpurun_out/pmc_sq.txt
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc_sq.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"]
print("per dispatch (sum over the chip); mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES x 4 SIMDs... see DESIGN) ; wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES")
print("kernel | dispatches | " + " | ".join(names) + " | wait% | issue-stall% | lds-conflict/lds-active")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    c = max(n[k].values())
    v = [d.get(x, 0.0) / max(1, n[k].get(x, 1)) for x in names]
    wc = v[0] or 1.0
    print(f"{k:64s} {c:5d} " + " ".join(f"{x:14.0f}" for x in v) + f"  {100*v[5]/wc:5.1f} {100*v[6]/wc:5.1f} {v[7]/max(v[4],1):7.4f}")
PY
head -n 14 $R/gpurun_out/pmc_sq.txt | cut -c1-260
