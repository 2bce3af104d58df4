#!/usr/bin/env python3
"""The 256 x 256 weight-gradient GEMMs of one 393 216-row chunk of a training step, alone on the GPU: one launch each (plain,
with the B2 rider, with the A2 rider), the six plain ones back to back on one stream, on two streams, and as ONE grouped launch
(swnerf_gemm_tn_group), with and without the two rider items in the group.  ms per call, TFLOP/s on the MFMA work."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib, model

dev = torch.device("cuda:0")
L = _lib.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
grad = torch.randn((M, 2432), device=dev)
act = torch.randn((M, 2432), device=dev)
xs = torch.randn((M, 96), device=dev)
d_out = torch.randn((M, 4), device=dev)
C = [torch.zeros((256, 319), device=dev) for _ in range(8)]
bias = [torch.zeros(256, device=dev) for _ in range(8)]
c5s, C3, b3 = torch.zeros((256, 64), device=dev), torch.zeros((1, 256), device=dev), torch.zeros(1, device=dev)
main = torch.cuda.current_stream(dev)
st = _lib.stream_of(grad)
FL = 2 * 256 * 256 * M


def plain(s, l):
    model._gemm_tn(L, s, M, grad, 256 * l, 256, act, 256 * (l - 1), 256, C[l], 0, bias[l])


def l5(s):
    model._gemm_tn_fused(L, s, M, grad, 1280, act, 1024, C[5], 63, bias[5], B2=xs, b2_col=0, Ni2=64, C2=c5s, c2_col=0)


def feat(s):
    model._gemm_tn_fused(L, s, M, grad, 2048, act, 1792, C[0], 0, bias[0], A2=d_out, a2_col=3, No2=1, C3=C3, bias3=b3)


def six_serial():
    for l in (1, 2, 3, 4, 6, 7):
        plain(st, l)


def six_two_streams():
    fan = model._Fan(dev)
    fan.fork()
    for l in (1, 2, 3, 4, 6, 7):
        plain(fan, l)
    fan.join()


def eight_two_streams():
    fan = model._Fan(dev)
    fan.fork()
    for l in (1, 2, 3, 4, 6, 7):
        plain(fan, l)
    l5(fan); feat(fan)
    fan.join()


def group(n_rider):
    def f():
        g = model._Group(st)
        for l in (1, 2, 3, 4, 6, 7):
            plain(g, l)
        if n_rider:
            l5(g); feat(g)
        g.launch(L, M)
    return f


def timeit(name, f, flops):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    while n < 10 or time.perf_counter() - t0 < 0.3:
        f()
        n += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"| {name} | {dt * 1e3:.3f} | {flops / dt / 1e12:.1f} |")


print(f"M = {M} rows;  SWNERF_GG_RIDER_W={os.environ.get('SWNERF_GG_RIDER_W', '(5)')}")
print("| what | ms | TFLOP/s (main 256x256 MFMA work only) |")
print("|---|---|---|")
timeit("one plain GEMM", lambda: plain(st, 1), FL)
timeit("one GEMM with the B2 rider (pts_linears.5)", lambda: l5(st), FL)
timeit("one GEMM with the A2 rider (feature_linear + alpha_linear)", lambda: feat(st), FL)
timeit("six plain, one stream", six_serial, 6 * FL)
timeit("six plain, two side streams", six_two_streams, 6 * FL)
timeit("six plain, ONE grouped launch", group(0), 6 * FL)
timeit("six plain + two with riders, two side streams", eight_two_streams, 8 * FL)
timeit("six plain + two with riders, ONE grouped launch", group(2), 8 * FL)
