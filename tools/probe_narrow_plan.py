#!/usr/bin/env python3
"""The table-driven narrow weight-gradient kernels (swnerf_deform_narrow_grads / swnerf_noview_narrow_grads) alone on the chip at
the chunk sizes of the fused backward passes, against the separate skinny GEMM launches they replace (SWNERF_NARROW_FUSED=0),
and swnerf_canon_narrow_grads for comparison: us per chunk, GB/s on the bytes read once.   usage: probe_narrow_plan.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib, wgrad

dev = torch.device("cuda:0")
L = _lib.lib()
st = _lib.stream_of(torch.empty(1, device=dev))
p = _lib.ptr


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


print("| product set | rows | fused kernel us | GB/s (operands once) | separate GEMMs us |")
print("|---|---|---|---|---|")
for kind, M in (("deform", 196608), ("noview", 393216), ("canon", 393216)):
    grad, act = torch.randn((M, 2432), device=dev), torch.randn((M, 2432), device=dev)
    xs = torch.randn((M, 96), device=dev)
    z = lambda *s: torch.zeros(s, device=dev)
    if kind == "deform":
        draw = torch.randn((M, 4), device=dev)
        c0s, cts, w4, b0, b4 = z(256, 64), z(256, 32), z(4, 256), z(256), z(4)
        fused = lambda: _lib.check(L.swnerf_deform_narrow_grads(p(grad), 2432, p(act), 2432, p(xs), p(draw), M, p(c0s), p(cts), p(w4), p(b0), p(b4), st), "dn")
        sep = lambda: (wgrad._gemm_tn(L, st, M, grad, 0, 256, xs, 0, 64, c0s, 0, b0), wgrad._gemm_tn(L, st, M, grad, 0, 256, xs, 64, 32, cts, 0, None),
                       wgrad._gemm_tn(L, st, M, draw, 0, 4, act, 1792, 256, w4, 0, b4))
        nbytes = M * 4 * (256 + 96 + 256 + 4)
    elif kind == "noview":
        draw = torch.randn((M, 8), device=dev)
        c0s, w8, b0, b8 = z(256, 64), z(8, 256), z(256), z(8)
        fused = lambda: _lib.check(L.swnerf_noview_narrow_grads(p(grad), 2432, p(act), 2432, p(xs), p(draw), M, p(c0s), p(w8), p(b0), p(b8), st), "nv")
        sep = lambda: (wgrad._gemm_tn(L, st, M, grad, 0, 256, xs, 0, 64, c0s, 0, b0), wgrad._gemm_tn(L, st, M, draw, 0, 8, act, 1792, 256, w8, 0, b8))
        nbytes = M * 4 * (256 + 96 + 256 + 8)
    else:
        draw = torch.randn((M, 4), device=dev)
        c0s, cvs, G, a4w, r4w, b0, bh, a4b, r4b = z(256, 64), z(128, 32), z(128, 256), z(4, 256), z(4, 128), z(256), z(128), z(4), z(4)
        fused = lambda: _lib.check(L.swnerf_canon_narrow_grads(p(grad), 2432, p(act), 2432, p(xs), p(draw), M, p(c0s), p(cvs), p(G), p(a4w), p(r4w), p(b0), p(bh),
                                                               p(a4b), p(r4b), st), "cn")
        sep = None
        nbytes = M * 4 * (256 + 256 + 128 + 128 + 96 + 4)
    tf = timed(fused)
    ts = timed(sep) if sep else float("nan")
    print(f"| {kind} | {M} | {tf:.0f} | {nbytes / tf / 1e3:.0f} | {ts:.0f} |")
    del grad, act, xs
