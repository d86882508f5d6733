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
        agg[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"# kernel trace: {os.path.relpath(f, d)}")
    print(f"{'calls':>7} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'total_ms':>10}  kernel")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if flt and flt not in k:
            continue
        print(f"{len(v):7d} {sum(v)/len(v):10.3f} {min(v):10.3f} {max(v):10.3f} {sum(v)/1e3:10.3f}  {k[:110]}")
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
