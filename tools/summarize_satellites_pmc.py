#!/usr/bin/env python3
"""Per-row PMC table of the satellite kernels from tools/profile_r04_satellites.sh's per-row runs: kernel duration, HBM bytes
from the counters (FETCH_SIZE x 2 for the 16-byte streaming reads - the gfx950 correction of MI355X_MICROARCH.md - plus
WRITE_SIZE) against the algorithmic bytes, GB/s on both, and how busy the vector ALUs were (VALUBusy = 4 x
SQ_ACTIVE_INST_VALU / 1024 SIMDs / GRBM_GUI_ACTIVE per XCD): these kernels stop being HBM bound when their instruction
count per byte is high (a wave64 VALU instruction occupies its SIMD for 4 cycles).  The counter tallies issue quad-cycles and
an f64 or transcendental instruction takes more than one, so a saturated kernel can read above 100 %: shown as ">= 100 %".
usage: summarize_satellites_pmc.py dir"""
import glob
import json
import os
import re
import sys

root = sys.argv[1]
KERNELS = ("get_rays_kernel", "pack_rays_kernel", "embed_kernel", "raw2outputs_kernel", "raw2outputs_bwd_kernel", "sample_pdf_kernel")
print("| row (tools/bench_satellites.py) | kernel us (rocprofv3) | algorithmic MB | counter MB (2 x FETCH + WRITE) | GB/s algorithmic | of 8 TB/s | of 6.29 TB/s | VALUBusy |")
print("|---|---|---|---|---|---|---|---|")
for d in sorted(glob.glob(os.path.join(root, "row*")), key=lambda p: int(re.sub(r"\D", "", os.path.basename(p)))):
    if not os.path.isdir(d):
        continue
    name = open(os.path.join(d, "name.txt")).read().strip()
    line = open(os.path.join(d, "line.txt")).read()
    pm = json.load(open(os.path.join(d, "pmc.json")))
    key = [k for k in pm if any(k.strip().startswith(n) or (" " + n) in k or n in k for n in KERNELS)]
    key = [k for k in key if "bwd" in k] if "backward" in name else [k for k in key if "bwd" not in k]
    if not key:
        continue
    k = pm[key[0]]
    mb_alg = float(re.search(r"us\s+([\d.]+) MB", line).group(1)) if re.search(r"us\s+([\d.]+) MB", line) else float("nan")
    us = k["avg_us"]
    mb_ctr = (2 * k.get("FETCH_SIZE", 0) + k.get("WRITE_SIZE", 0)) * 1024 / 1e6
    gbs = mb_alg * 1e6 / (us * 1e-6) / 1e9
    valu = 4 * k.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / (k.get("GRBM_GUI_ACTIVE", 1) / 8)
    print(f"| {name} | {us:.1f} | {mb_alg:.1f} | {mb_ctr:.1f} | {gbs:.0f} | {gbs / 8000:.1%} | {gbs / 6290:.1%} | {'>= 100 %' if valu >= 1 else format(valu, '.0%')} |")
