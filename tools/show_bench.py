#!/usr/bin/env python3
"""Print a bench.py record (the file holding its JSON line, possibly behind other stdout lines) as a table."""
import json
import sys

line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
d = json.loads(line)
print({k: v for k, v in d.items() if k not in ("extra", "config")})
for r in d.get("extra", {}).get("configs", []):
    fr = r["frac"]
    print(f"{r['name'][:72]:72s} {r['ms']:8.3f} ms {r['rays_per_s']:10.0f} r/s alg {r['algorithmic_tflops']:6.1f} exec {r.get('executed_tflops', 0):6.1f} "
          f"frac {fr if fr is None else round(fr, 4)} psnr {r.get('psnr_vs_cpu_render_db')}")
