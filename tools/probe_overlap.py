#!/usr/bin/env python3
"""Do an MFMA-bound pass kernel and the HBM-heavy weight-gradient GEMMs run faster side by side (two streams) than one
after the other?  Stand-ins: the fine render pass (4096 rays x 192) and 9 swnerf_gemm_tn launches at 393 216 rows."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib, synth, model, render

dev = torch.device("cuda:0")
L = _lib.lib()
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
K, c2w = synth.lego_camera(800, 800)
o, d = synth.pick_rays(800, 800, K, c2w, 4096, 2)
rb = render.pack_ray_batch(torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev), 2., 6.)
z = torch.linspace(2, 6, 192, device=dev).expand(4096, 192).contiguous()
M = 393216
A = torch.randn((M, 2432), device=dev)
B = torch.randn((M, 2432), device=dev)
C = torch.zeros((9, 256, 256), device=dev)
s_main, s_side = torch.cuda.Stream(), torch.cuda.Stream()


def pass_on(st):
    with torch.cuda.stream(st), torch.no_grad():
        render.render_pass(rb, net, 192, z_vals=z, white_bkgd=True)


def gemms_on(st):
    with torch.cuda.stream(st):
        for l in range(9):
            _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * 256 * (l % 8), 2432, 256, B.data_ptr() + 4 * 256 * ((l + 1) % 8), 2432, 256, M,
                                        C[l].data_ptr(), 256, None, st.cuda_stream), "g")


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.default_stream())
    s_main.wait_stream(torch.cuda.default_stream()); s_side.wait_stream(torch.cuda.default_stream())
    for _ in range(reps):
        fn()
    torch.cuda.default_stream().wait_stream(s_main); torch.cuda.default_stream().wait_stream(s_side)
    e1.record(torch.cuda.default_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("| schedule | ms per (fine pass + 9 GEMMs) |")
print("|---|---|")
print(f"| pass alone | {timed(lambda: pass_on(s_main)):.3f} |")
print(f"| 9 GEMMs alone | {timed(lambda: gemms_on(s_main)):.3f} |")
print(f"| one stream: pass, then GEMMs | {timed(lambda: (pass_on(s_main), gemms_on(s_main))):.3f} |")
print(f"| two streams: pass || GEMMs | {timed(lambda: (pass_on(s_main), gemms_on(s_side))):.3f} |")
print(f"| two streams: GEMMs issued first | {timed(lambda: (gemms_on(s_side), pass_on(s_main))):.3f} |")
