#!/usr/bin/env python3
"""Marginal rate of the 256 x 256 weight-gradient GEMM (swnerf_gemm_tn on 16-byte aligned operands = gemm_tn_dma_kernel)
with the operands served from the Infinity Cache (footprint < 256 MB, launched repeatedly) against HBM (footprint >> 256 MB):
time(2M) - time(M) removes the fixed cost of a launch (ramp + atomic epilogue).  Answers: is the kernel's 78 % the matrix
side (then the cache-resident rate is no better) or the memory side?"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/sw-nerf_amd')
import torch
from swnerf import _lib
L = _lib.lib()
dev = torch.device('cuda:0')


def t_of(M, ld, reps=20):
    A = torch.randn((M, ld), device=dev)
    B = torch.randn((M, ld), device=dev)
    C = torch.zeros((256, 256), device=dev)
    bias = torch.zeros(256, device=dev)
    st = _lib.stream_of(A)
    f = lambda: _lib.check(L.swnerf_gemm_tn(A.data_ptr(), ld, 256, B.data_ptr(), ld, 256, M, C.data_ptr(), 256, bias.data_ptr(), st), 'g')
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


print("| operands | M | ms | marginal TFLOP/s (from the row above) | footprint |")
print("|---|---|---|---|---|")
for ld, what in ((256, "dense [M,256]"), (2432, "windows of [M,2432]")):
    prev = None
    for M in (24576, 49152, 98304, 196608, 393216, 786432):
        if ld == 2432 and M > 393216:
            continue
        t = t_of(M, ld)
        marg = "" if prev is None else f"{2 * (M - prev[0]) * 65536 / (t - prev[1]) / 1e12:.1f}"
        print(f"| {what} | {M} | {t * 1e3:.3f} | {marg} | {M * 2048 / 2 ** 20:.0f} MB touched |")
        prev = (M, t)
