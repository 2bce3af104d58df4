#!/usr/bin/env python3
"""Per kernel name (prefix-trimmed) and counter: average over dispatches of the counter value summed over instances,
plus the average dispatch duration.  usage: summarize_pmc_any.py dir_with_pmc_subdirs [name_filter]"""
import csv
import glob
import sys
from collections import defaultdict

flt = sys.argv[2] if len(sys.argv) > 2 else ""
print("| kernel | counter | avg per dispatch | dispatches | avg us |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/**/*_counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    dur = {}
    names = {}
    for r in csv.DictReader(open(f)):
        if flt and flt not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        names[d] = r["Kernel_Name"].split("(")[0][-48:] + f" grid{r['Grid_Size']}"
    agg = defaultdict(lambda: defaultdict(list))
    for d in per:
        for c, v in per[d].items():
            agg[names[d]][c].append((v, dur[d]))
    for k in sorted(agg):
        for c in sorted(agg[k]):
            vs = agg[k][c]
            print(f"| {k} | {c} | {sum(v for v, _ in vs) / len(vs):.6g} | {len(vs)} | {sum(t for _, t in vs) / len(vs) / 1e3:.1f} |")
