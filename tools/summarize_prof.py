#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel trace / counter collection) into a
small text summary for profiles/.   python tools/summarize_prof.py <dir> [name-filter]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    agg = defaultdict(list)
    for r in csv.DictReader(open(f)):
        # one line per (kernel, grid): the same kernel serves several workloads (e.g. the vec-mat of C2 and of the 14336^2 shape)
        key = (r["Kernel_Name"], int(r.get("Grid_Size_X", 0)) // max(int(r.get("Workgroup_Size_X", 1)), 1))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# kernel trace: {os.path.relpath(f, d)}")
    print(f"{'calls':>7} {'avg_us':>10} {'median_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10} {'workgroups':>10}  kernel")
    for (k, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if flt and flt not in k:
            continue
        v = sorted(v)
        print(f"{len(v):7d} {sum(v)/len(v):10.3f} {v[len(v)//2]:10.3f} {v[0]:10.3f} {v[-1]:10.3f} {sum(v)/1e3:10.3f} {g:10d}  {k[:100]}")
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"# counters: {os.path.relpath(f, d)}  (mean per dispatch)")
    for k, cs in agg.items():
        if flt and flt not in k:
            continue
        print(f"  {k[:110]}")
        for c, v in sorted(cs.items()):
            print(f"    {c:34s} {sum(v)/len(v):18.1f}   (n={len(v)})")
