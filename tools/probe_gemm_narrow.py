#!/usr/bin/env python3
"""The skinny weight-gradient GEMMs of a training step at the chunk size the fused backward uses (393 216 rows):
per-launch time and the HBM rate their operand reads amount to.  SWNERF_GEMM_NARROW_OLD=1 selects the round-2 kernel
(single-buffered, two workgroups per CU) for the same shapes."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/sw-nerf_amd')
import torch
from swnerf import _lib
L = _lib.lib()
dev = torch.device('cuda:0')
M = int(sys.argv[1]) if len(sys.argv) > 1 else 393216
grad = torch.randn((M, 2432), device=dev)
act = torch.randn((M, 2432), device=dev)
xs = torch.randn((M, 96), device=dev)
draw = torch.randn((M, 4), device=dev)
st = _lib.stream_of(grad)
print(f"SWNERF_GEMM_NARROW_OLD={os.environ.get('SWNERF_GEMM_NARROW_OLD', '(unset)')}  M={M}")
print("| GEMM | No x Ni | us | TFLOP/s | operand bytes / time |")
print("|---|---|---|---|---|")
for name, A, a_col, No, B, b_col, Ni in (
        ("pts_linears.0 (gamma(x) slots)", grad, 0, 256, xs, 0, 64), ("views_linears.0 x feature", grad, 2304, 128, act, 2048, 256),
        ("views_linears.0 x gamma(d) slots", grad, 2304, 128, xs, 64, 32), ("rgb_linear (4-row form)", draw, 0, 4, act, 2304, 128),
        ("trunk layer (wide, for scale)", grad, 256, 256, act, 0, 256)):
    C = torch.zeros((No, Ni), device=dev)
    bias = torch.zeros(No, device=dev)
    f = lambda: _lib.check(L.swnerf_gemm_tn(A.data_ptr() + 4 * a_col, A.stride(0), No, B.data_ptr() + 4 * b_col, B.stride(0), Ni, M,
                                            C.data_ptr(), Ni, bias.data_ptr(), st), "g")
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"| {name} | {No} x {Ni} | {dt * 1e6:.1f} | {2 * M * No * Ni / dt / 1e12:.1f} | {M * (No + Ni) * 4 / dt / 1e12:.2f} TB/s |")
