#!/usr/bin/env python3
"""One training step of the reference (nerf/run.py:635-708) on the differentiable op path, C2 shape:
4096 rays, 64 coarse + 128 fine samples, coarse + fine nets, loss = mse(rgb) + mse(rgb0), backward, Adam.
Secondary metric (the headline bench.py is the forward render); prints a markdown table + per-kernel times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, embedder, render

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
embed_fn, c10 = embedder.get_embedder(10, 3, 0)
embeddirs_fn, c4 = embedder.get_embedder(4, 3, 0)
nets = []
for seed, ab in (synth.NET_COARSE, synth.NET_FINE):
    m = model.vallina_NeRF(D=8, W=256, input_ch=c10, input_ch_views=c4, output_ch=5, skips=[4], use_viewdirs=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(seed, alpha_bias=ab).items()})
    nets.append(m.to(dev))
q = lambda inputs, viewdirs, network_fn: render.run_network(inputs, viewdirs, network_fn, embed_fn=embed_fn,
                                                            embeddirs_fn=embeddirs_fn, netchunk=1024 * 64)
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, N, 2)
rays = (torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev))
target = torch.rand((N, 3), device=dev)
opt = torch.optim.Adam([p for m in nets for p in m.parameters()], lr=5e-4, betas=(0.9, 0.999))
img2mse = lambda x, y: torch.mean((x - y) ** 2)
kw = dict(ndc=False, near=2., far=6., use_viewdirs=True, network_fn=nets[0], network_query_fn=q, N_samples=64, N_importance=128,
          network_fine=nets[1], white_bkgd=True, perturb=1., raw_noise_std=0.)


def step(timers=None):
    ev = lambda: torch.cuda.Event(enable_timing=True)
    e = [ev() for _ in range(4)]
    e[0].record()
    rgb, disp, acc, extras = render.render(800, 800, K, chunk=1024 * 32, rays=rays, **kw)
    loss = img2mse(rgb, target) + img2mse(extras['rgb0'], target)
    e[1].record()
    opt.zero_grad()
    loss.backward()
    e[2].record()
    opt.step()
    e[3].record()
    if timers is not None:
        torch.cuda.synchronize()
        timers.append([e[i].elapsed_time(e[i + 1]) for i in range(3)])
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
timers = []
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    step(timers)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
t = np.mean(np.array(timers), 0)
flop = N * 256 * 2 * (527872 + 495616 + 527872)   # EXECUTED: forward + dX chain + dW, feature_linear folded (bench.py FLOP_EXEC_TRAIN_PER_ROW)
print(f"| training step, {N} rays x (64+128), two nets, fp32 | {dt*1e3:.1f} ms/step | {N/dt:,.0f} rays/s | forward {t[0]:.1f} ms, backward {t[1]:.1f} ms, Adam {t[2]:.1f} ms | ~{flop/dt/1e12:.0f} TFLOP/s |")
print(f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
