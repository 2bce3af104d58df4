#!/usr/bin/env python3
"""The bf16x3 / bf16 render pass against the fp32 pass on the C2 scene: deviation of the raw MLP outputs and of the
composited image, and time per launch.  usage: probe_x3.py [n_rays]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import _lib
if os.environ.get("SWNERF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SWNERF_LIB"])
from swnerf import synth, model, render

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")


def load(spec):
    net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(spec[0], alpha_bias=spec[1]).items()})
    return net.to(dev).eval()


coarse, fine = load(synth.NET_COARSE), load(synth.NET_FINE)
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, N, 2)
rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)


def psnr(a, b):
    return float(-10.0 * torch.log10(torch.mean((a.double() - b.double()) ** 2)))


def timed(f, reps=10):
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    want = ("rgb_map", "disp_map", "acc_map", "raw")
    c32 = render.render_pass(rb, coarse, 64, white_bkgd=True, want=want, n_importance=128, precision="fp32")
    z = c32["z_fine"]
    f32 = render.render_pass(rb, fine, 192, z_vals=z, white_bkgd=True, want=want, precision="fp32")
    print(f"{N} rays, 64 + 128 samples; fp32 pass = reference of this table")
    print("| precision | max abs d raw (coarse) | max abs d raw (fine, same depths) | max abs d rgb (fine, same depths) | PSNR fine rgb vs fp32, same depths (dB) | "
          "PSNR end to end (own coarse pass + resampling) | coarse ms | fine ms | rays/s (coarse + fine) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for prec in ("fp32", "bf16x3", "bf16"):
        c = render.render_pass(rb, coarse, 64, white_bkgd=True, want=want, n_importance=128, precision=prec)
        f = render.render_pass(rb, fine, 192, z_vals=z, white_bkgd=True, want=want, precision=prec)
        e = render.render_pass(rb, fine, 192, z_vals=c["z_fine"], white_bkgd=True, want=want, precision=prec)
        tc = timed(lambda: render.render_pass(rb, coarse, 64, white_bkgd=True, n_importance=128, precision=prec))
        tf = timed(lambda: render.render_pass(rb, fine, 192, z_vals=z, white_bkgd=True, precision=prec))
        print(f"| {prec} | {float((c['raw'] - c32['raw']).abs().max()):.3e} | {float((f['raw'] - f32['raw']).abs().max()):.3e} | "
              f"{float((f['rgb_map'] - f32['rgb_map']).abs().max()):.3e} | {psnr(f['rgb_map'], f32['rgb_map']):.1f} | "
              f"{psnr(e['rgb_map'], f32['rgb_map']):.1f} | {tc:.3f} | {tf:.3f} | {N / (tc + tf) * 1e3:,.0f} |")
        if prec == "fp32":
            print(f"|   (scale) max abs raw = {float(f32['raw'].abs().max()):.2f}, mean acc = {float(f32['acc_map'].mean()):.3f} | | | | | | | | |")
