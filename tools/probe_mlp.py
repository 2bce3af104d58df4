#!/usr/bin/env python3
"""Kernel-level timing of the MLP entry points at the C2 fine-pass row count (M = 4096 x 192 = 786 432):
inference forward, training forward (saves activations + ReLU bit masks), dX chain, and the 14 weight-gradient GEMMs.
usage: probe_mlp.py [lib.so]   (an alternative build for timing experiments)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "sw-nerf_amd"), ROOT):
    sys.path.insert(0, p)
import torch
from swnerf import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from swnerf import synth, model

dev = torch.device("cuda:0")
M = 4096 * 192
FLOP_ROW = 2 * (593408 - 65536)      # executed: feature_linear folded into the view layer
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(*synth.NET_FINE[:1], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev)
x = torch.rand((M, 90), device=dev) * 2 - 1
G = torch.randn((M, 4), device=dev)


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    t_inf = timed(lambda: net(x))
L = _lib.lib()
kind, packed, Lp, Ld, _ = net.packed()
out = torch.empty((M, 4), device=dev)
act = torch.empty((M, L.swnerf_act_floats_per_row()), device=dev)
bits = torch.empty(L.swnerf_mask_floats(M), device=dev)
grad = torch.empty_like(act)
st = _lib.stream_of(x)
t_fwd = timed(lambda: _lib.check(L.swnerf_mlp_forward_train(_lib.ptr(packed), _lib.ptr(x), M, Lp, Ld, _lib.ptr(out), _lib.ptr(act), _lib.ptr(bits), st), "fwd"))
pb = net.packed_bwd()
t_bwd = timed(lambda: _lib.check(L.swnerf_mlp_backward_dx(_lib.ptr(pb), _lib.ptr(bits), _lib.ptr(G), M, _lib.ptr(grad), st), "bwd"))
sd_ = dict(net.named_parameters())
wg = model.WeightGrads(L, "canon", [sd_[n] for n in model._CANON_ORDER], fused=False, Cpos=63, Cdir=27, bands=(10, 4, 0))
t_gemm = timed(lambda: wg.chunk(st, M, grad, act, x, G))
tf = lambda ms, f: M * f / (ms * 1e-3) / 1e12
print(f"| M = {M} rows | ms | algorithmic TFLOP/s | % of 157.3 |")
print("|---|---|---|---|")
for name, ms, f in (("mlp_forward (inference)", t_inf, FLOP_ROW), ("mlp_forward_train", t_fwd, FLOP_ROW),
                    ("mlp_backward_dx", t_bwd, 2 * 495616), ("weight-gradient GEMMs (14 launches)", t_gemm, FLOP_ROW)):
    print(f"| {name} | {ms:.2f} | {tf(ms, f):.1f} | {100 * tf(ms, f) / 157.3:.1f} |")
