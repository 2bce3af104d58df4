#!/usr/bin/env python3
"""bf16x3 against fp32 on the C5 shard (D-NeRF, t = 0.5), equal depths: distribution of the per-ray deviation."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import numpy as np
import torch
from swnerf import synth, model, render, embedder, ray as swray

dev = torch.device("cuda:0")
e10, _ = embedder.get_embedder(10, 3, 0)
dn = model.NeRF.get_by_name("direct_temporal", D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27,
                            input_ch_time=21, use_viewdirs=True, embed_fn=e10, zero_canonical=True)
dn.load_state_dict({k: torch.from_numpy(v) for k, v in synth.dnerf_state_dict(synth.NET_DNERF[0], alpha_bias=synth.NET_DNERF[1]).items()})
dn = dn.to(dev).eval()
K, c2w = synth.lego_camera(400, 400)
lo, hi = synth.shard_range(400 * 400, 8, 3)
o, d = swray.get_rays(400, 400, torch.from_numpy(np.asarray(K, dtype=np.float32)), torch.from_numpy(np.asarray(c2w, dtype=np.float32)).to(dev))
o, d = o.reshape(-1, 3)[lo:hi].contiguous(), d.reshape(-1, 3)[lo:hi].contiguous()
rb = render.pack_ray_batch(o, d, 2., 6., frame_time=0.5)
want = ("rgb_map", "acc_map", "raw", "dx")
with torch.no_grad():
    c = render.render_pass(rb, dn, 64, white_bkgd=True, n_importance=128, want=(), precision="fp32")
    a = render.render_pass(rb, dn, 192, z_vals=c["z_fine"], white_bkgd=True, want=want, precision="fp32")
    b = render.render_pass(rb, dn, 192, z_vals=c["z_fine"], white_bkgd=True, want=want, precision="bf16x3")
    b2 = render.render_pass(rb, dn, 192, z_vals=c["z_fine"], white_bkgd=True, want=want, precision="bf16x3")
print("repeatable:", torch.equal(b["rgb_map"], b2["rgb_map"]))
err = (a["rgb_map"] - b["rgb_map"]).abs().max(dim=1).values.cpu().numpy()
print("rays", err.size, "max", err.max(), "p50", np.percentile(err, 50), "p99", np.percentile(err, 99), "p99.9", np.percentile(err, 99.9))
print("psnr", float(-10 * torch.log10(torch.mean((a["rgb_map"].double() - b["rgb_map"].double()) ** 2))))
bad = np.nonzero(err > 1e-3)[0]
print("rays with |d rgb| > 1e-3:", bad.size, bad[:40], "mod 4:", np.bincount(bad % 4, minlength=4) if bad.size else None)
ddx = (a["dx"] - b["dx"]).abs().amax(dim=(1, 2)).cpu().numpy()
draw = (a["raw"] - b["raw"]).abs().amax(dim=(1, 2)).cpu().numpy()
print("max |d dx| per ray: max", ddx.max(), "p99", np.percentile(ddx, 99), "; max |d raw| per ray: max", draw.max(), "p99", np.percentile(draw, 99))
if bad.size:
    i = int(bad[np.argmax(err[bad])])
    print("worst ray", i, "d rgb", err[i], "d dx", ddx[i], "d raw", draw[i], "acc", float(a["acc_map"][i]), float(b["acc_map"][i]))
    s = int((a["raw"][i] - b["raw"][i]).abs().amax(dim=1).argmax())
    print(" worst sample", s, "raw fp32", a["raw"][i, s].cpu().numpy(), "x3", b["raw"][i, s].cpu().numpy(), "dx fp32", a["dx"][i, s].cpu().numpy(), "x3", b["dx"][i, s].cpu().numpy())
