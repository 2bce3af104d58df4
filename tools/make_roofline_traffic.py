#!/usr/bin/env python3
"""profiles/roofline_traffic.json from THIS round's PMC passes: HBM/fabric bytes per fine-pass launch of the headline
(C2) workload = FETCH_SIZE x 2 (gfx950 tallies the 128-B requests of wide streaming reads at 64 B: MI355X_MICROARCH.md, HBM)
+ WRITE_SIZE, both reported in KB per dispatch summed over the XCDs.  bench.py puts the figure and this file's round /
source into its JSON line.   usage: make_roofline_traffic.py pmc_summary.json round "source text" """
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
fine = [k for k in d if "render_pass_kernel<false, false" in k and k.endswith("[fine]")]
coarse = [k for k in d if "render_pass_kernel<false, false" in k and k.endswith("[coarse]")]
assert len(fine) == 1 and len(coarse) == 1, (fine, coarse)
f, c = d[fine[0]], d[coarse[0]]
out = {
    "round": int(sys.argv[2]),
    "source": sys.argv[3],
    "fine_pass_FETCH_SIZE_KB_reported": f["FETCH_SIZE"], "fine_pass_WRITE_SIZE_KB_reported": f["WRITE_SIZE"],
    "coarse_pass_FETCH_SIZE_KB_reported": c["FETCH_SIZE"], "coarse_pass_WRITE_SIZE_KB_reported": c["WRITE_SIZE"],
    "fine_pass_launches_averaged": f["dispatches"],
    "correction": "gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) streaming reads -> doubled (MI355X_MICROARCH.md, HBM)",
    "fine_pass_hbm_bytes_per_launch": int(round(2 * f["FETCH_SIZE"] * 1024 + f["WRITE_SIZE"] * 1024)),
    "coarse_pass_hbm_bytes_per_launch": int(round(2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024)),
}
json.dump(out, open(os.path.join(ROOT, "profiles", "roofline_traffic.json"), "w"), indent=2)
print(json.dumps(out, indent=2))
