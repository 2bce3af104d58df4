#!/usr/bin/env python3
"""Per-launch time of the fused render pass at given depths per ray (the fine-pass form: z_vals given, no resampling)
for several sample counts - separates per-ray from per-tile cost.  usage: probe_pass.py [lib.so]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from swnerf import synth, model, render

dev = torch.device("cuda:0")
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, 4096, 2)
rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
print("| samples per ray | tiles | ms per launch (4096 rays) |")
print("|---|---|---|")
with torch.no_grad():
    for S in (32, 64, 96, 192, 384):
        z = torch.linspace(2, 6, S, device=dev).expand(4096, S).contiguous()
        f = lambda: render.render_pass(rb, net, S, z_vals=z, white_bkgd=True)
        f(); f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"| {S} | {S // 32} | {e0.elapsed_time(e1) / 20:.4f} |")

    print("\n| coarse form (depths computed in the kernel), 64 samples | ms per launch (4096 rays) |")
    print("|---|---|")
    for name, kw in (("no resampling", dict()), ("+ sample_pdf/merge, 128 fine samples", dict(n_importance=128)),
                     ("+ the same with random u (sort-first branch)", dict(n_importance=128, u=torch.rand((4096, 128), device=dev)))):
        f = lambda: render.render_pass(rb, net, 64, white_bkgd=True, **kw)
        f(); f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        print(f"| {name} | {e0.elapsed_time(e1) / 20:.4f} |")
