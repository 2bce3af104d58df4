#!/usr/bin/env python3
"""Throughput of the generic (layer-by-layer) path: a use_viewdirs=False 8x256 net on [M,63] embedded rows, and the
end-to-end render (4096 rays, 64+128) without view directions, against the fused path with view directions."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import synth, model, embedder, render

dev = torch.device("cuda:0")
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False).to(dev).eval()
M = 262144
x = torch.randn((M, 63), device=dev)
with torch.no_grad():
    net(x); net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        net(x)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
macs = 63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 * 5
print(f"| generic MLP forward, {M} rows | {dt * 1e3:.2f} ms | {2 * macs * M / dt / 1e12:.1f} TFLOP/s |")
embed_fn, _ = embedder.get_embedder(10, 3, 0)
embeddirs_fn = None
q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, 4096, 2)
rays = (torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
kw = dict(ndc=False, near=2., far=6., use_viewdirs=False, network_fn=net, network_query_fn=q, N_samples=64, N_importance=128, network_fine=None,
          white_bkgd=True, perturb=0., raw_noise_std=0.)
with torch.no_grad():
    render.render(800, 800, K, rays=rays, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        render.render(800, 800, K, rays=rays, **kw)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"| render(), 4096 rays x (64+128), use_viewdirs=False (generic path) | {dt * 1e3:.1f} ms | {4096 / dt:,.0f} rays/s |")
