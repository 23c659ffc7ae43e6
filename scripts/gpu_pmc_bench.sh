#!/bin/bash
# HBM traffic of the bench step per kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE pmc passes (TCC slots),
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  Output: gpurun_out/pmc_traffic.txt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_$C -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-extras > $R/gpurun_out/pmc_$C.log 2>&1
  find /tmp/pmc_$C -name "*counter_collection*.csv" -exec cp {} /tmp/pmc_$C.csv \;
done
python3 - <<PY > $R/gpurun_out/pmc_traffic.txt
import csv, collections
out = collections.defaultdict(dict)
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    rows = list(csv.DictReader(open(f"/tmp/pmc_{C}.csv")))
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in rows:
        if r["Counter_Name"] != C: continue
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
    for k, (v, n) in agg.items(): out[k][C] = (v, n)
print("kernel | dispatches | FETCH_SIZE KiB/dispatch (raw counter) | WRITE_SIZE KiB/dispatch | note: gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x")
tot_f = tot_w = 0
for k, d in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", (0, 1))[0]):
    f, n = d.get("FETCH_SIZE", (0, 1)); w, n2 = d.get("WRITE_SIZE", (0, 1))
    print(f"{k:70s} {n:6d} {f/max(n,1):14.1f} {w/max(n2,1):14.1f}")
PY
head -n 25 $R/gpurun_out/pmc_traffic.txt
