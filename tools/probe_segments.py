#!/usr/bin/env python3
"""Where do the cycles of a tile go?  Runs the fine pass (4096 rays x 192 samples, C2 shape) on a -DSW_PROBE build of the
library (tools/experiments/probe/build.sh): every wave stamps the shader clock around sampling + encoding, the 8-layer trunk
(+ sigma head), the feature / view / rgb tail and the compositing of each tile, and around its whole tile loop.
Prints per-tile averages against the ideal matrix-pipe time of each part (4 MFMAs x 64 cycles per weight step)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import _lib
_lib.LIB_PATH = os.environ.get("SWNERF_PROBE_LIB", os.path.join(ROOT, "tools", "experiments", "probe", "libswnerf_probe.so"))
from swnerf import synth, model, render

dev = torch.device("cuda:0")
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, 4096, 2)
rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
COARSE = len(sys.argv) > 1 and sys.argv[1] == "coarse"          # the coarse pass of C2: 64 samples + the resampling tail (128 draws)
S = 64 if COARSE else 192
z = None if COARSE else torch.linspace(2, 6, S, device=dev).expand(4096, S).contiguous()
kw = dict(n_importance=128) if COARSE else {}
with torch.no_grad():
    for _ in range(3):
        out = render.render_pass(rb, net, S, z_vals=z, white_bkgd=True, want=["rgb_map", "weights"], **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = render.render_pass(rb, net, S, z_vals=z, white_bkgd=True, want=["rgb_map", "weights"], **kw)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
allw = out["weights"].cpu().numpy().view(np.uint64).reshape(4096, -1).astype(np.float64)
raw = allw[:, :6]
resample_cycles = allw[:, 9] if COARSE else None
ntiles = S // 32
per_tile = raw[:, :4] / ntiles
loop, prologue = raw[:, 4], raw[:, 5]
ideal = {"trunk (8 layers + ReLU + sigma head)": (64 + 4 * 256 + 256 + 64 + 2 * 256) * 256, "tail (view layer on h7 - feature_linear folded in, gamma(d) once per ray -, rgb head)": 128 * 256}
names = ["sampling + gamma(x)", "trunk (8 layers + ReLU + sigma head)", "tail (view layer on h7 - feature_linear folded in, gamma(d) once per ray -, rgb head)", "compositing (+ tile-loop bookkeeping)"]
print(f"{'coarse' if COARSE else 'fine'} pass, probe build: {ms:.3f} ms per launch (4096 rays x {S} samples; the stamps add their own s_memtime + waits)")
print()
print("| part of a 32-sample tile | cycles (mean over 4096 waves) | ideal matrix-pipe cycles | excess | share of the tile |")
print("|---|---|---|---|---|")
tot = per_tile.sum(1).mean()
for i, n in enumerate(names):
    c = per_tile[:, i].mean()
    idl = ideal.get(n, 0)
    print(f"| {n} | {c:,.0f} | {idl:,} | {c - idl:,.0f} | {100 * c / tot:.2f} % |")
print(f"| whole tile | {tot:,.0f} | {sum(ideal.values()):,} | {tot - sum(ideal.values()):,.0f} | 100 % |")
print()
print(f"tile loop per wave: {loop.mean():,.0f} cycles = {ntiles} x {loop.mean() / ntiles:,.0f}; prologue (kernel entry -> first tile: bias copy, barrier, "
      f"pe_dir, ring prime): {prologue.mean():,.0f} cycles (min {prologue.min():,.0f}, max {prologue.max():,.0f})")
print(f"slowest / fastest wave tile loop: {loop.max():,.0f} / {loop.min():,.0f} cycles; matrix pipe busy if only the MFMAs counted: "
      f"{100 * sum(ideal.values()) / tot:.2f} %")
if COARSE:
    print(f"resampling tail (sample_pdf, z_std, rank merge; matrix pipe idle): {resample_cycles.mean():,.0f} cycles per ray (min {resample_cycles.min():,.0f}, "
          f"max {resample_cycles.max():,.0f}) = {100 * resample_cycles.mean() / (loop.mean() + prologue.mean() + resample_cycles.mean()):.2f} % of the wave's time")
