#!/usr/bin/env python3
"""Launch times of the fused pass in the single-round regime (1024 rays = one wave per SIMD on 256 CUs) and at 4096 rays,
for the three forms a render uses: coarse 64 samples (C1), coarse + resampling, fine 192 samples on given depths; plus
render_rays end to end at the north_star shape (1024 rays x (64+128), two nets).
The start-up shaping of the kernel is chosen per PROCESS by SWNERF_WARM / SWNERF_SKEW (csrc/render_pass.h
pass_startup_args), so run this once per variant:   SWNERF_WARM=0 SWNERF_SKEW=0 python tools/probe_small_batch.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from swnerf import synth, model, render, embedder

dev = torch.device("cuda:0")
nets = []
for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
    m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed, alpha_bias=ab).items()})
    nets.append(m.to(dev).eval())
embed_fn, _ = embedder.get_embedder(10, 3, 0)
embeddirs_fn, _ = embedder.get_embedder(4, 3, 0)
query = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn)
K, c2w = synth.lego_camera(800, 800)
FLOP_ROW = 2 * 593408
PEAK = 157.3e12


def timed(f, reps=50):
    f(); f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3          # us


print(f"SWNERF_WARM={os.environ.get('SWNERF_WARM', '(default)')} SWNERF_SKEW={os.environ.get('SWNERF_SKEW', '(default)')}")
print("| rays | form | us per launch | fraction of 157.3 TFLOP/s |")
print("|---|---|---|---|")
with torch.no_grad():
    for N in (1024, 2048, 4096):
        o, d = synth.pick_rays(800, 800, K, c2w, N, 2)
        rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
        z = torch.linspace(2, 6, 192, device=dev).expand(N, 192).contiguous()
        for name, rows, f in (
                ("coarse 64, no resampling (C1)", 64, lambda: render.render_pass(rb, nets[0], 64, white_bkgd=True)),
                ("coarse 64 + resampling to 192", 64, lambda: render.render_pass(rb, nets[0], 64, white_bkgd=True, n_importance=128)),
                ("fine 192 on given depths", 192, lambda: render.render_pass(rb, nets[1], 192, z_vals=z, white_bkgd=True)),
                ("render_rays 64+128, two nets", 256, lambda: render.render_rays(rb, nets[0], query, 64, N_importance=128, network_fine=nets[1], white_bkgd=True))):
            us = timed(f, 50 if N == 1024 else 20)
            print(f"| {N} | {name} | {us:.1f} | {N * rows * FLOP_ROW / (us * 1e-6) / PEAK:.4f} |")
