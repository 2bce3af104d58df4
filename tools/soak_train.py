#!/usr/bin/env python3
"""Training soak: a randomly initialised coarse+fine pair (swnerf.runner.create_nerf, the reference's train() loop:
render -> mse(rgb)+mse(rgb0) -> backward -> Adam -> lr decay, nerf/run.py:635-708) learns to reproduce the renders of a
fixed "teacher" pair (the synthetic bench nets) from random rays of 8 cameras.  Prints the PSNR against the teacher on
held-out rays every 100 steps.  usage: soak_train.py [steps=600] [N_rand=1024]"""
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, runner, render

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
args = SimpleNamespace(expname="soak", basedir="/tmp/swnerf_soak", netdepth=8, netwidth=256, netdepth_fine=8, netwidth_fine=256,
                       lrate=5e-4, lrate_decay=250, netchunk=1024 * 64, no_reload=True, ft_path=None, N_samples=64, N_importance=128,
                       perturb=1., use_viewdirs=True, i_embed=0, multires=10, multires_views=4, raw_noise_std=0.,
                       dataset_type="blender", white_bkgd=True, no_ndc=False, lindisp=False)
torch.manual_seed(0)
train_kw, test_kw, _, grad_vars, opt = runner.create_nerf(args, device=dev)
teacher = []
for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
    m = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed, alpha_bias=ab).items()})
    teacher.append(m.to(dev).eval())
tkw = dict(test_kw, network_fn=teacher[0], network_fine=teacher[1])
H = W = 200
cams = [synth.lego_camera(H, W, theta=th) for th in np.linspace(-180, 180, 9)[:-1]]
K = cams[0][0]
rng = np.random.default_rng(0)


def batch(n, seed):
    os_, ds_ = [], []
    for ci, (Kc, c2w) in enumerate(cams):
        o, d = synth.pick_rays(H, W, Kc, c2w, n // len(cams), seed * 100 + ci)
        os_.append(o); ds_.append(d)
    return torch.from_numpy(np.concatenate(os_)).to(dev), torch.from_numpy(np.concatenate(ds_)).to(dev)


def teach(rays):
    with torch.no_grad():
        return render.render(H, W, K, chunk=1024 * 32, rays=rays, near=2., far=6., **tkw)[0]


held = batch(2048, 9999)
held_t = teach(held)
img2mse = lambda x, y: torch.mean((x - y) ** 2)
psnr = lambda: float(-10 * torch.log10(img2mse(render.render(H, W, K, chunk=1024 * 32, rays=held, near=2., far=6., **test_kw)[0], held_t)))
print(f"| step | loss | PSNR vs teacher on held-out rays (dB) |")
print("|---|---|---|")
with torch.no_grad():
    print(f"| 0 | - | {psnr():.2f} |")
t0 = time.perf_counter()
for i in range(steps):
    rays = batch(N, i)
    target = teach(rays)
    rgb, disp, acc, extras = render.render(H, W, K, chunk=1024 * 32, rays=rays, near=2., far=6., **train_kw)
    opt.zero_grad()
    loss = img2mse(rgb, target) + img2mse(extras['rgb0'], target)
    loss.backward()
    opt.step()
    for pg in opt.param_groups:
        pg['lr'] = args.lrate * (0.1 ** ((i + 1) / (args.lrate_decay * 1000)))
    if (i + 1) % 100 == 0:
        with torch.no_grad():
            print(f"| {i + 1} | {float(loss.detach()):.5f} | {psnr():.2f} |", flush=True)
torch.cuda.synchronize()
print(f"{steps} steps of {N} rays in {time.perf_counter() - t0:.1f} s (incl. the teacher renders and the PSNR probes)")
