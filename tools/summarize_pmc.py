#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for the render kernel: per counter, the
average over the FINE-pass dispatches (render() launches coarse then fine alternately).
usage: summarize_pmc.py dir_with_pmc_subdirs"""
import csv
import glob
import sys
from collections import defaultdict

out = {}
for f in sorted(glob.glob(sys.argv[1] + "/pmc_*/runc/*_counter_collection.csv")):
    per = defaultdict(lambda: defaultdict(float))     # dispatch -> counter -> value (summed over instances)
    dur = {}
    for r in csv.DictReader(open(f)):
        if "render_pass_kernel" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    ids = sorted(per)
    fine = ids[1::2][1:]       # alternate launches, drop the first fine pass (warm-up)
    coarse = ids[0::2][1:]
    for name in sorted({c for d in per.values() for c in d}):
        out[name] = (sum(per[d][name] for d in fine) / len(fine), sum(per[d][name] for d in coarse) / len(coarse),
                     sum(dur[d] for d in fine) / len(fine) / 1e3, len(fine))
print("| counter | fine pass (avg/launch) | coarse pass (avg/launch) | fine launch us (profiled) | launches |")
print("|---|---|---|---|---|")
for k, v in out.items():
    print(f"| {k} | {v[0]:.6g} | {v[1]:.6g} | {v[2]:.1f} | {v[3]} |")
