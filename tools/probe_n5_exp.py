#!/usr/bin/env python3
"""What bounds narrow5_kernel (the five narrow weight-gradient products of a 393 216-row chunk)?  Experiment builds of
csrc/backward_kernels.hip (tools/experiments/gemm/build.sh <name> -DN5_EXP_NOBARRIER | -DN5_EXP_NODMA | -DN5_EXP_NOMFMA: WRONG results,
timing only) against the shipped loop.  usage: probe_n5_exp.py libgemm_a.so [libgemm_b.so ...]"""
import ctypes
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib

dev = torch.device("cuda:0")
_lib.lib()
M = 393216
grad, act = torch.randn((M, 2432), device=dev), torch.randn((M, 2432), device=dev)
xs, draw = torch.randn((M, 96), device=dev), torch.randn((M, 4), device=dev)
z = lambda *s: torch.zeros(s, device=dev)
c0s, cvs, G, a4w, rgb4, b0, bhv, a4b, rgb4b = z(256, 64), z(128, 32), z(128, 256), z(4, 256), z(4, 128), z(256), z(128), z(4), z(4)
st = _lib.stream_of(grad)
P = ctypes.c_void_p
print("| build | narrow5, 393 216 rows: us | TFLOP/s (52 992 MAC/row) | operand GB/s (3 472 B/row) |")
print("|---|---|---|---|")
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    fn = L.swnerf_canon_narrow_grads
    fn.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P, P, ctypes.c_int64] + [P] * 10
    f = lambda: fn(grad.data_ptr(), 2432, act.data_ptr(), 2432, xs.data_ptr(), draw.data_ptr(), M, c0s.data_ptr(), cvs.data_ptr(), G.data_ptr(),
                   a4w.data_ptr(), rgb4.data_ptr(), b0.data_ptr(), bhv.data_ptr(), a4b.data_ptr(), rgb4b.data_ptr(), st)
    for _ in range(3):
        assert f() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"| {os.path.basename(path)} | {us:.1f} | {2 * 52992 * M / us / 1e6:.1f} | {3472 * M / us / 1e3:.0f} |")
