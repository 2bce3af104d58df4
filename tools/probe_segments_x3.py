#!/usr/bin/env python3
"""Where do the cycles of a bf16x3 tile go?  The fine pass (4096 rays x 192 samples) on a -DSW_PROBE build:
  tools/experiments/x3/build.sh probe_plain -DSW_PROBE -DX3_NO_PIPE   the plain form (split phase between layers), stamped inside:
      MFMA segments (ring waits and barriers included) | accumulator -> (hi, lo) splits + heads | gamma(x) | rest
  tools/experiments/x3/build.sh probe_pipe -DSW_PROBE                 the shipped software-pipelined form: whole-MLP cycles only
usage: [SWNERF_LIB=...so] probe_segments_x3.py [bf16x3|bf16]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import _lib
_lib.LIB_PATH = os.path.abspath(os.environ.get("SWNERF_LIB", os.path.join(ROOT, "tools", "experiments", "x3", "libswnerf_probe_plain.so")))
from swnerf import synth, model, render

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
terms = 3 if prec == "bf16x3" else 1
dev = torch.device("cuda:0")
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, 4096, 2)
rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
S = 192
z = torch.linspace(2, 6, S, device=dev).expand(4096, S).contiguous()
with torch.no_grad():
    for _ in range(3):
        out = render.render_pass(rb, net, S, z_vals=z, white_bkgd=True, want=["rgb_map", "weights"], precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    out = render.render_pass(rb, net, S, z_vals=z, white_bkgd=True, want=["rgb_map", "weights"], precision=prec)
    e1.record()
    torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
raw = out["weights"].cpu().numpy().view(np.uint64).reshape(4096, -1)[:, :9].astype(np.float64)
nt = S // 32
groups = 1160
ideal = groups * terms * 32
front, mlp, comp, loop, prologue = raw[:, 0] / nt, raw[:, 1] / nt, raw[:, 3] / nt, raw[:, 4] / nt, raw[:, 5]
seg, split, pe = raw[:, 6] / nt, raw[:, 7] / nt, raw[:, 8] / nt
print(f"{prec} fine pass, probe build: {ms:.3f} ms per launch (4096 rays x {S} samples; the stamps add their own s_memtime + waits)\n")
print("| part of a 32-sample tile | cycles (mean over 4096 waves) | note |")
print("|---|---|---|")
stamped = seg.mean() > 0
if not stamped:
    print(f"| the MLP call (software-pipelined: no stamps inside) | {mlp.mean():,.0f} | ideal {ideal:,} MFMA cycles |")
if stamped: print(f"| MFMA segments | {seg.mean():,.0f} | ideal {ideal:,} = {groups} groups x {terms} MFMA x 32 cycles; the excess {seg.mean() - ideal:,.0f} is ring waits, barriers, operand reads, bias reads |")
if stamped:
    print(f"| accumulators -> relu -> (hi, lo) + heads | {split.mean():,.0f} | 9 layers |")
    print(f"| gamma(x), twice, + its split | {pe.mean():,.0f} | |")
    print(f"| rest of the MLP call | {(mlp - seg - split - pe).mean():,.0f} | stamps, loop control |")
print(f"| sampling | {front.mean():,.0f} | |")
print(f"| compositing (+ tile-loop bookkeeping) | {comp.mean():,.0f} | |")
print(f"| whole tile | {loop.mean():,.0f} | matrix pipe busy if only the MFMAs counted: {100 * ideal / loop.mean():.1f} % |")
print(f"\nprologue: {prologue.mean():,.0f} cycles; slowest / fastest wave tile loop: {raw[:, 4].max():,.0f} / {raw[:, 4].min():,.0f} cycles")
