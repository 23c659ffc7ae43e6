"""Per-dispatch view of a rocprofv3 --kernel-trace CSV: for every (kernel, grid) class the durations, the idle gap in
front of it and the kernel that ran before it; and the ordered launch list of one steady-state step.

usage: python scripts/trace_gaps.py <kernel_trace.csv> [launches_per_step]
"""
import csv, sys, collections, re

path = sys.argv[1]
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0),
                     int(r.get("LDS_Block_Size", 0) or 0), int(r.get("VGPR_Count", 0) or 0)))
rows.sort()


def short(n):
    n = re.sub(r"\(.*$", "", n)
    n = n.replace("void ", "")
    return n[:60]


agg = collections.OrderedDict()
prev_end, prev_name = None, "-"
for (s, e, n, g, w, lds, vg) in rows:
    key = (short(n), g // max(w, 1))
    a = agg.setdefault(key, dict(n=0, dur=0.0, gap=0.0, durs=[], prevs=collections.Counter(), lds=lds, vg=vg))
    a["n"] += 1
    a["dur"] += (e - s) / 1e3
    a["durs"].append((e - s) / 1e3)
    if prev_end is not None:
        a["gap"] += max(0, s - prev_end) / 1e3
    a["prevs"][short(prev_name)] += 1
    prev_end, prev_name = e, n

tot = sum(a["dur"] for a in agg.values())
totgap = sum(a["gap"] for a in agg.values())
print(f"{len(rows)} dispatches, kernel time {tot/1e3:.3f} ms, idle gaps {totgap/1e3:.3f} ms")
print(f"{'kernel':62s} {'blocks':>7s} {'n':>5s} {'avg us':>8s} {'min':>7s} {'p50':>7s} {'max':>7s} {'gap us':>7s} {'sum ms':>8s}  LDS  VGPR  most frequent predecessor")
for (name, blocks), a in sorted(agg.items(), key=lambda kv: -kv[1]["dur"]):
    d = sorted(a["durs"])
    print(f"{name:62s} {blocks:7d} {a['n']:5d} {a['dur']/a['n']:8.2f} {d[0]:7.2f} {d[len(d)//2]:7.2f} {d[-1]:7.2f} {a['gap']/a['n']:7.2f} {a['dur']/1e3:8.3f}  {a['lds']:5d} {a['vg']:4d}  {a['prevs'].most_common(1)[0][0][:40]}")

if per_step:
    # the last complete step: ordered list with start offsets
    step = rows[-per_step:]
    t0 = step[0][0]
    print(f"\nlast {per_step} dispatches (one step), offsets in us:")
    pe = None
    for (s, e, n, g, w, lds, vg) in step:
        gap = (s - pe) / 1e3 if pe is not None else 0.0
        print(f"  +{(s-t0)/1e3:9.2f}  dur {(e-s)/1e3:8.2f}  gap {gap:6.2f}  blocks {g//max(w,1):6d}  {short(n)}")
        pe = e
