#!/bin/bash
# vL1D (TCP) counters per kernel of the bench step (own --pmc pass, kernel-trace only): read requests, L1 hits, requests
# passed on to the L2 and their accumulated latency.  Output: gpurun_out/pmc_tcp.txt
set -u
R="${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_tcp
timeout 900 rocprofv3 --pmc TCP_PERF_SEL_TOTAL_READ TCP_PERF_SEL_TOTAL_HIT_LRU_READ TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY --kernel-trace --output-format csv -d /tmp/pmc_tcp -o r -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-decode --no-extras > $R/gpurun_out/pmc_tcp.log 2>&1
find /tmp/pmc_tcp -name "*counter_collection*.csv" -exec cp {} /tmp/pmc_tcp.csv \;
python3 - <<PY > $R/gpurun_out/pmc_tcp.txt
import csv, collections
rows = list(csv.DictReader(open("/tmp/pmc_tcp.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.Counter())
for r in rows:
    k = r["Kernel_Name"].split("(")[0][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
names = ["TCP_PERF_SEL_TOTAL_READ", "TCP_PERF_SEL_TOTAL_HIT_LRU_READ", "TCP_TCC_READ_REQ", "TCP_TCC_READ_REQ_LATENCY"]
print("per dispatch (sum over the chip)")
print("kernel | dispatches | " + " | ".join(names) + " | L1 hit % | L2 requests per L1 read | mean L2 read latency (cycles)")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("TCP_TCC_READ_REQ", 0)):
    c = max(n[k].values())
    v = [d.get(x, 0.0) / max(1, n[k].get(x, 1)) for x in names]
    print(f"{k:64s} {c:5d} " + " ".join(f"{x:16.0f}" for x in v) + f"  {100*v[1]/max(v[0],1):5.1f} {v[2]/max(v[0],1):6.3f} {v[3]/max(v[2],1):8.1f}")
PY
head -n 16 $R/gpurun_out/pmc_tcp.txt | cut -c1-260
