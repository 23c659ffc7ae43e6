#!/bin/bash
# One --pmc pass (kernel-trace only) over the bench step with an arbitrary counter list; per-kernel averages per dispatch.
#   scripts/gpu_pmc_generic.sh <name> "<COUNTER ...>"   ->  gpurun_out/pmc_<name>.txt
# Also joins the kernel trace of the SAME pass: mean duration per dispatch, so ratios like GRBM_GUI_ACTIVE / wall (the
# effective shader clock, MI355X_MICROARCH.md "DVFS give-back") come from one run.
set -u
NAME=$1; COUNTERS=$2
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$NAME
timeout 900 rocprofv3 --pmc $COUNTERS --kernel-trace --output-format csv -d /tmp/pmc_$NAME -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-extras > $R/gpurun_out/pmc_$NAME.log 2>&1
find /tmp/pmc_$NAME -name "*counter_collection*.csv" -exec cp {} /tmp/pmc_$NAME.csv \;
find /tmp/pmc_$NAME -name "*kernel_trace*.csv" -exec cp {} /tmp/pmc_${NAME}_trace.csv \;
NAME=$NAME COUNTERS="$COUNTERS" python3 - <<'PY' > $R/gpurun_out/pmc_$NAME.txt
import csv, collections, os, re
name, names = os.environ["NAME"], os.environ["COUNTERS"].split()
rows = list(csv.DictReader(open(f"/tmp/pmc_{name}.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
def short(k): return re.sub(r"\(.*$", "", k).replace("void ", "")[:64]
for r in rows:
    k = short(r["Kernel_Name"])
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(lambda: [0.0, 0])
try:
    for r in csv.DictReader(open(f"/tmp/pmc_{name}_trace.csv")):
        d = dur[short(r["Kernel_Name"])]; d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
except OSError:
    pass
print("per dispatch (sum over the chip), same-pass mean kernel duration in us")
print("kernel | dispatches | us | " + " | ".join(names))
for k, d in sorted(agg.items(), key=lambda kv: -dur[kv[0]][0]):
    c = max(n[k].values())
    us = dur[k][0] / max(1, dur[k][1])
    print(f"{k:64s} {c:5d} {us:9.2f} " + " ".join(f"{d.get(x, 0.0) / max(1, n[k].get(x, 1)):16.0f}" for x in names))
PY
head -n 12 $R/gpurun_out/pmc_$NAME.txt | cut -c1-240
