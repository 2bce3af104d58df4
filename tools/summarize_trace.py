#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, LDS size, grid) average duration.
The coarse and the fine render pass are the same kernel launched alternately by render(); the
trace does not report dynamic LDS, so they are told apart by launch order.  usage: summarize_trace.py kernel_trace.csv [skip_first_n_per_group]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
g = defaultdict(list)
seq = defaultdict(int)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][-60:]
    if "render_pass_kernel" in name:      # render() launches coarse then fine, alternately
        name += " [coarse 64 smp]" if seq[name] % 2 == 0 else " [fine 192 smp]"
        seq[name.split(" [")[0]] += 1
    g[(name, r["LDS_Block_Size"], r["Grid_Size_X"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"])].append(
        int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | LDS B | grid.x | VGPR | AGPR | scratch | calls | avg us | min us | max us |")
print("|---|---|---|---|---|---|---|---|---|---|")
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v = v[skip:] if len(v) > skip else v
    print(f"| {k[0]} | {k[1]} | {k[2]} | {k[3]} | {k[4]} | {k[5]} | {len(v)} | {sum(v)/len(v)/1e3:.1f} | {min(v)/1e3:.1f} | {max(v)/1e3:.1f} |")
