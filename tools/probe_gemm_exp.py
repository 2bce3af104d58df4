#!/usr/bin/env python3
"""Where does the 256 x 256 weight-gradient GEMM group lose its 20 %?  The group launch of one 393 216-row chunk (six plain items +
the skip layer's with its gamma(x) rider, as a training step issues it) on experiment builds of the kernel
(tools/experiments/gemm/build.sh: no slab barrier / no refill DMA / no VALU work / no atomic epilogue - WRONG results, timing only).
usage: probe_gemm_exp.py libgemm_a.so [libgemm_b.so ...]"""
import ctypes
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib

dev = torch.device("cuda:0")
_lib.lib()
M = 393216
grad, act = torch.randn((M, 2432), device=dev), torch.randn((M, 2432), device=dev)
xs = torch.randn((M, 96), device=dev)
C = [torch.zeros((256, 320), device=dev) for _ in range(8)]
bias = [torch.zeros(256, device=dev) for _ in range(8)]
c5s = torch.zeros((256, 64), device=dev)
st = _lib.stream_of(grad)
items = []
for l in (1, 2, 3, 4, 6, 7):
    items.append(_lib.GemmItem(grad.data_ptr() + 4 * 256 * l, 2432, act.data_ptr() + 4 * 256 * (l - 1), 2432, C[l].data_ptr(), 320, bias[l].data_ptr(),
                               None, 0, 0, None, 0, None, 0, 0, None, 0, None))
plain = (_lib.GemmItem * 6)(*items)
items.append(_lib.GemmItem(grad.data_ptr() + 4 * 1280, 2432, act.data_ptr() + 4 * 1024, 2432, C[5].data_ptr() + 4 * 64, 320, bias[5].data_ptr(),
                           xs.data_ptr(), 96, 64, c5s.data_ptr(), 64, None, 0, 0, None, 0, None))
seven = (_lib.GemmItem * 7)(*items)
print("| build | six plain items: ms | TFLOP/s | + the rider item: ms | TFLOP/s |")
print("|---|---|---|---|---|")
for path in sys.argv[1:]:
    L = ctypes.CDLL(os.path.abspath(path))
    L.swnerf_gemm_tn_group.argtypes = [ctypes.POINTER(_lib.GemmItem), ctypes.c_int, ctypes.c_int64, ctypes.c_void_p]
    out = []
    for arr, n in ((plain, 6), (seven, 7)):
        f = lambda: L.swnerf_gemm_tn_group(arr, n, M, st)
        for _ in range(3):
            assert f() == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out += [f"{ms:.3f}", f"{n * 2 * 65536 * M / ms / 1e9:.1f}"]
    print(f"| {os.path.basename(path)} | " + " | ".join(out) + " |")
