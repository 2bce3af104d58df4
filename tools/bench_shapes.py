#!/usr/bin/env python3
"""render() in a loop at one of the shapes profiles/r03 documents, for rocprofv3 (kernel trace / PMC):
  north_star  lego 400x400 camera, 1024-ray batch x (64+128), two 8x256 nets with view directions (nerf/configs/lego.txt)
  noview      4096-ray batch x (64+128), two 8x256 nets WITHOUT view directions (use_viewdirs=False, the reference's default)
usage: bench_shapes.py north_star|noview [reps]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import synth, model, render, embedder

what = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
embed_fn, in_ch = embedder.get_embedder(10, 3, 0)
if what == "north_star":
    embeddirs_fn, in_views = embedder.get_embedder(4, 3, 0)
    specs = [(synth.nerf_state_dict(s, alpha_bias=ab), dict(input_ch_views=in_views, use_viewdirs=True)) for s, ab in (synth.NET_COARSE, synth.NET_FINE)]
    H = W = 400
    N, flop_row, use_views = 1024, 2 * (593408 - 65536 - 4096) + 2 * 2 * 131072 / 256, True       # EXECUTED MACs per row: feature_linear folded, gamma(d) once per ray and pass
else:
    embeddirs_fn = None
    specs = [(synth.noview_state_dict(s, alpha_bias=ab), dict(input_ch_views=0, use_viewdirs=False)) for s, ab in ((20250321, 0.5), (20250322, 0.7))]
    H = W = 800
    N, flop_row, use_views = 4096, 2 * (63 * 256 + 4 * 256 * 256 + 319 * 256 + 2 * 256 * 256 + 256 * 5), False
nets = []
for sd, kw in specs:
    m = model.vallina_NeRF(D=8, W=256, input_ch=in_ch, output_ch=5, skips=[4], **kw)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    nets.append(m.to(dev).eval())
q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
K, c2w = synth.lego_camera(H, W)
o, d = synth.pick_rays(H, W, K, c2w, N, 2)
rays = (torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
kw = dict(ndc=False, near=2., far=6., use_viewdirs=use_views, network_fn=nets[0], network_query_fn=q, N_samples=64, N_importance=128,
          network_fine=nets[1], white_bkgd=True, perturb=0., raw_noise_std=0.)
with torch.no_grad():
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        render.render(H, W, K, rays=rays, **kw)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        render.render(H, W, K, rays=rays, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
print(f"{what}: {N} rays x (64+128): {dt * 1e3:.3f} ms per render() = {N / dt:,.0f} rays/s = {N * 256 * flop_row / dt / 157.3e12:.4f} of 157.3 TFLOP/s on executed FLOPs "
      f"(wall clock incl. Python, {reps} reps; under a profiler the kernels are serialised with extra gaps)")
