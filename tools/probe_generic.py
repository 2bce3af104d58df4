#!/usr/bin/env python3
"""Throughput of the generic (layer-by-layer) path - the nets the register-resident kernels are not built for: forward of
a D=8 / W=256 / skips=[2,5] net with view directions and of a D=6 / W=384 net on embedded rows, their training step
(forward + backward through swnerf_linear / swnerf_gemm_nn / swnerf_gemm_tn), and the use_viewdirs=False 8x256 net's
training forward+backward.  SWNERF_GENERIC_GEMM_OLD=1 selects the round-2 64x64 GEMM kernel for comparison."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import model

dev = torch.device("cuda:0")
M = 262144
print(f"SWNERF_GENERIC_GEMM_OLD={os.environ.get('SWNERF_GENERIC_GEMM_OLD', '(unset)')}  M={M}")
print("| net | what | ms | TFLOP/s (2 x MACs x rows; x3 for a training step) |")
print("|---|---|---|---|")
for name, kw in (("D=8 W=256 skips=[2,5] use_viewdirs", dict(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[2, 5], use_viewdirs=True)),
                 ("D=6 W=384 skips=[3] use_viewdirs", dict(D=6, W=384, input_ch=63, input_ch_views=27, output_ch=5, skips=[3], use_viewdirs=True)),
                 ("D=8 W=256 skips=[4] use_viewdirs=False (training runs here)", dict(D=8, W=256, input_ch=63, input_ch_views=0, output_ch=5, skips=[4], use_viewdirs=False))):
    net = model.vallina_NeRF(**kw).to(dev)
    x = torch.randn((M, kw["input_ch"] + kw["input_ch_views"]), device=dev)
    W, D = kw["W"], kw["D"]
    macs = kw["input_ch"] * W + sum((W + kw["input_ch"] if (i - 1) in kw["skips"] else W) * W for i in range(1, D))
    macs += (W * W + W + (W + kw["input_ch_views"]) * (W // 2) + (W // 2) * 3) if kw["use_viewdirs"] else W * kw["output_ch"]
    for what, grad in (("forward", False), ("forward + backward", True)):
        if not grad and not net._is_fused_arch() and net._noview_params() is not None:
            continue                                  # that forward runs on the register-resident kernel
        net.train(grad)

        def f():
            if grad:
                for p_ in net.parameters():
                    p_.grad = None
                net(x).sum().backward()
            else:
                with torch.no_grad():
                    net(x)
        f(); f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        print(f"| {name} | {what} | {dt * 1e3:.2f} | {(3 if grad else 1) * 2 * macs * M / dt / 1e12:.1f} |")
