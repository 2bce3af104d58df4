#!/usr/bin/env python3
"""Per kernel name (prefix-trimmed) and counter: average over dispatches of the counter value summed over instances,
plus the average dispatch duration.  Dispatches of the fused render pass are split by LAUNCH ORDER into its coarse
(64 samples) and fine (192 samples) launches - render() issues them alternately with the same kernel name and grid - as
tools/summarize_trace.py does for the kernel trace; the first `skip` launches of each are dropped (warm-up).
usage: summarize_pmc_any.py dir_with_pmc_subdirs [name_filter] [--skip N] [--json out.json]
--json: additionally write {kernel label: {counter: avg, "dispatches": n, "avg_us": t}} for tools/make_roofline_traffic.py"""
import csv
import glob
import json
import sys
from collections import defaultdict

args = [a for a in sys.argv[1:]]
skip, jpath = 0, None
if "--skip" in args:
    i = args.index("--skip"); skip = int(args[i + 1]); del args[i:i + 2]
if "--json" in args:
    i = args.index("--json"); jpath = args[i + 1]; del args[i:i + 2]
root = args[0]
flt = args[1] if len(args) > 1 else ""
dump = defaultdict(dict)
print("| kernel | counter | avg per dispatch | dispatches | avg us |")
print("|---|---|---|---|---|")
for f in sorted(glob.glob(root + "/pmc_*/**/*_counter_collection.csv", recursive=True)):
    per = defaultdict(lambda: defaultdict(float))
    dur, names = {}, {}
    for r in csv.DictReader(open(f)):
        if flt and flt not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[d] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        names[d] = r["Kernel_Name"].split("(")[0][-48:] + f" grid{r['Grid_Size']}"
    seq = defaultdict(int)
    label = {}
    for d in sorted(per):                              # dispatch ids follow launch order
        n = names[d]
        if "render_pass_kernel" in n and "backward" not in n:
            n += " [coarse]" if seq[names[d]] % 2 == 0 else " [fine]"
            seq[names[d]] += 1
        label[d] = n
    agg = defaultdict(lambda: defaultdict(list))
    for d in sorted(per):
        for c, v in per[d].items():
            agg[label[d]][c].append((v, dur[d]))
    for k in sorted(agg):
        for c in sorted(agg[k]):
            vs = agg[k][c]
            vs = vs[skip:] if len(vs) > skip else vs
            avg, us = sum(v for v, _ in vs) / len(vs), sum(t for _, t in vs) / len(vs) / 1e3
            print(f"| {k} | {c} | {avg:.6g} | {len(vs)} | {us:.1f} |")
            dump[k][c] = avg
            dump[k]["dispatches"], dump[k]["avg_us"] = len(vs), us
if jpath:
    json.dump(dump, open(jpath, "w"), indent=1, sort_keys=True)
