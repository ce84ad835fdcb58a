#!/usr/bin/env python3
"""Per-launch means of every counter collected by tools/pmc_stall.sh for the production traversal kernel,
split into the primary and the bounce launches (they alternate), plus the launch durations."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))   # counter -> dispatch id -> values
order = {}
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(path)) if "k_traverse_wide<float, 10" in r["Kernel_Name"]]
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    rank = {d: k for k, d in enumerate(ids)}
    for r in rows:
        vals[r["Counter_Name"]][rank[int(r["Dispatch_Id"])]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(path)) if "k_traverse_wide<float, 10" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # dispatch order of tools/pmc_traffic.py: one host-path primary launch, then (primary, bounce) pairs
    for k, r in enumerate(rows):
        if k >= 1:
            dur[(k - 1) % 2].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in (0, 1):
    if dur[k]:
        print("%s launch: %.1f us mean under the profiler (%d launches)" % (("primary", "bounce")[k], sum(dur[k]) / len(dur[k]), len(dur[k])))
print("%-40s %16s %16s" % ("counter", "primary", "bounce"))
for c in sorted(vals):
    p = [sum(v) for d, v in vals[c].items() if d % 2 == 1]
    b = [sum(v) for d, v in vals[c].items() if d % 2 == 0 and d >= 2]
    if p and b:
        print("%-40s %16.0f %16.0f" % (c, sum(p) / len(p), sum(b) / len(b)))
