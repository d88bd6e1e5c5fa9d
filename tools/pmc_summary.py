#!/usr/bin/env python3
"""Summarise tools/pmc.sh output: mean counter value per kernel."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:34]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in sorted(glob.glob(sys.argv[1] + "/g*/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:34]
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
want = sys.argv[2:] or None
for name, c in agg.items():
    if want and not any(w in name for w in want):
        continue
    d = sorted(dur[name])
    print(f"{name}: n={len(d)} median {d[len(d)//2]:.1f} us")
    for k, v in sorted(c.items()):
        print(f"    {k:38s} {sum(v)/len(v):14.4g}")
