#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) average duration.
The coarse and the fine render pass are the same kernel launched alternately by render(); they are told apart by launch
order.  rocprofv3's VGPR / AGPR / LDS columns are NOT reproduced: for these kernels they are wrong (it prints VGPR 224 /
AGPR 0 / LDS 0 for a kernel that holds 444 registers, 192 of them AGPRs, and 136 KB of dynamic LDS); the real numbers
come from the code-object metadata: profiles/rNN/kernel_resources.md (tools/kernel_resources.py).
usage: summarize_trace.py kernel_trace.csv [skip_first_n_per_group]"""
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
        key = (name, r["Grid_Size_X"])
        name += " [coarse 64 smp]" if seq[key] % 2 == 0 else " [fine 192 smp]"
        seq[key] += 1
    g[(name, r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | grid.x | calls | avg us | min us | max us |")
print("|---|---|---|---|---|---|")
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    v = v[skip:] if len(v) > skip else v
    print(f"| {k[0]} | {k[1]} | {len(v)} | {sum(v)/len(v)/1e3:.1f} | {min(v)/1e3:.1f} | {max(v)/1e3:.1f} |")
