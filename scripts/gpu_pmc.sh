#!/bin/bash
# PMC pass over a command: usage gpu_pmc.sh "<counters>" <tag> -- cmd...
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out
CNT="$1"; TAG="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout 900 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d /tmp/pmc_$TAG -o r -- "$@" > $R/gpurun_out/pmc_$TAG.log 2>&1
find /tmp/pmc_$TAG -name "*counter_collection*.csv" -exec cp {} $R/gpurun_out/pmc_$TAG.csv \;
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$R/gpurun_out/pmc_$TAG.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    n = sum(1 for r in rows if r["Kernel_Name"][:60] == k and r["Counter_Name"] == list(d)[0])
    print(k, "dispatches", n)
    for c, v in d.items(): print(f"    {c:32s} {v/n:16.1f} per dispatch")
PY
