#!/usr/bin/env python3
"""One D-NeRF training step (d_nerf/run_dnerf.py:686-735) on the differentiable op path, bouncingballs-like shape:
N rays (default 4096; the shipped config uses N_rand=500), 64 coarse + 128 fine samples, ONE DirectTemporalNeRF for
both passes (the coarse pass runs under no_grad, run_dnerf.py:417-421), loss = mse(rgb) + tv_weight * TV(position_delta)
against a second render at a neighbouring time on the same depths, backward, Adam."""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, runner, render_dnerf

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
args = SimpleNamespace(expname="bench", basedir="/tmp/swnerf_bench", netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                       lrate=5e-4, netchunk=1024 * 64, no_reload=True, ft_path=None, N_samples=64, N_importance=128, perturb=1.,
                       use_viewdirs=True, i_embed=0, multires=10, multires_views=4, raw_noise_std=0., dataset_type="blender",
                       white_bkgd=True, no_ndc=False, lindisp=False, nerf_type="direct_temporal", not_zero_canonical=False,
                       use_two_models_for_fine=False, do_half_precision=False)
train_kw, _, _, grad_vars, opt = runner.create_dnerf(args, device=dev)
train_kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in synth.dnerf_state_dict(synth.NET_DNERF[0], alpha_bias=synth.NET_DNERF[1]).items()})
K, c2w = synth.lego_camera(400, 400)
o, d = synth.pick_rays(400, 400, K, c2w, N, 5)
rays = torch.stack([torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)], 0)
target = torch.rand((N, 3), device=dev)
img2mse = lambda x, y: torch.mean((x - y) ** 2)
focal = float(K[0, 0])


def step(tv, timers=None):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    ev[0].record()
    rgb, disp, acc, extras = render_dnerf.render(400, 400, focal, chunk=1024 * 32, rays=rays, frame_time=0.5, near=2., far=6.,
                                                 retraw=True, **train_kw)
    loss = img2mse(rgb, target)
    if tv:
        _, _, _, ex2 = render_dnerf.render(400, 400, focal, chunk=1024 * 32, rays=rays, frame_time=0.47, near=2., far=6.,
                                           retraw=True, z_vals=extras['z_vals'].detach(), **train_kw)
        loss = loss + 0.1 * (extras['position_delta'] - ex2['position_delta']).pow(2).sum() / N
    ev[1].record()
    opt.zero_grad()
    loss.backward()
    ev[2].record()
    opt.step()
    ev[3].record()
    if timers is not None:
        torch.cuda.synchronize()
        timers.append([ev[i].elapsed_time(ev[i + 1]) for i in range(3)])


print("| D-NeRF training step (t = 0.5), fp32, 1x MI355X | ms/step | rays/s | forward / backward / Adam ms |")
print("|---|---|---|---|")
for tv in ((False,) if "notv" in sys.argv else (False, True)):      # `notv`: the image-loss step alone (for rocprofv3 timelines)
    for _ in range(2):
        step(tv)
    torch.cuda.synchronize()
    timers, reps = [], 4
    t0 = time.perf_counter()
    for _ in range(reps):
        step(tv, timers)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    t = np.mean(np.array(timers), 0)
    print(f"| {N} rays x (64+128){' + TV loss (second render at a neighbouring time)' if tv else ''} | {dt*1e3:.1f} | {N/dt:,.0f} | {t[0]:.1f} / {t[1]:.1f} / {t[2]:.1f} |")
print(f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
