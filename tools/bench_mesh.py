#!/usr/bin/env python3
"""Throughput of the mesh-extraction grid query (nerf/extract_mesh.py sample_grid: resolution^3 points x
100 view directions through model_fine) on one MI355X: fused shared-direction kernel vs the reference's
loop structure (one full network query per view direction) on the same kernels."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, mesh

dev = torch.device("cuda:0")
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
R, V = 128, 100
bounds = [(-1., 1.), (-1., 2.), (-4., 2.)]
ax = [np.linspace(b[0], b[1], R) for b in bounds]
X, Y, Z = np.meshgrid(*ax, indexing="ij")
pts = torch.tensor(np.stack([X.ravel(), Y.ravel(), Z.ravel()], -1), dtype=torch.float32, device=dev)
dirs = torch.tensor(mesh.generate_viewdirs(V), dtype=torch.float32, device=dev)
M = pts.shape[0]
with torch.no_grad():
    mesh.query_points(net, pts[:4096], dirs, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = mesh.query_points(net, pts, dirs, True)
    torch.cuda.synchronize()
    t_fused = time.perf_counter() - t0
    sub = pts[:M // 16]                                # the per-view loop on 1/16 of the grid, scaled
    t0 = time.perf_counter()
    acc = torch.zeros((sub.shape[0], 4), device=dev)
    for v in range(V):
        acc += mesh.query_points(net, sub, dirs[v:v + 1].expand(sub.shape[0], 3).contiguous(), False)
    torch.cuda.synchronize()
    t_loop = (time.perf_counter() - t0) * 16
    err = float((acc[:, :3] / V - out[:M // 16, :3]).abs().max())
flop_fused = M * 2 * (593408 - 20480 - 18432 + V * (18432 + 128 * 3 + 283 * 128 - 18432))   # informational only
print(f"| grid {R}^3 = {M:,} points x {V} view directions | fused shared-direction query | {t_fused*1e3:.0f} ms | {M*V/t_fused/1e6:.0f} M point-views/s |")
print(f"| same, one full network query per direction (the reference's loop, extrapolated from 1/16 of the grid) | | {t_loop*1e3:.0f} ms | {M*V/t_loop/1e6:.0f} M point-views/s |")
print(f"| max abs difference of the view-averaged colours between the two | {err:.2e} | | |")
