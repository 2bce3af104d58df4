#!/usr/bin/env python3
"""mlp_forward16 (16-row tiles, 2 waves per SIMD) vs mlp_forward (32-row tiles, 1 wave per SIMD): same outputs? time?"""
import ctypes
import os
import sys
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, ROOT + '/sw-nerf_amd')
import torch
from swnerf import _lib, synth, model

dev = torch.device("cuda:0")
_lib.lib()                                   # the shipped library (32-row kernels), torch's HIP runtime first
L = ctypes.CDLL(os.path.join(HERE, "libtile16.so"))
L.tile16_packed_floats.restype = ctypes.c_size_t
L.tile16_pack_net.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.tile16_mlp_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
net = model.vallina_NeRF(D=8, W=256, input_ch=63, input_ch_views=27, output_ch=5, skips=[4], use_viewdirs=True)
net.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state_dict(synth.NET_FINE[0], alpha_bias=synth.NET_FINE[1]).items()})
net = net.to(dev).eval()
sd = dict(net.named_parameters())
ps = [sd[n].detach().contiguous() for n in model._CANON_ORDER]
arr = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in ps])
p16 = torch.empty(L.tile16_packed_floats(), dtype=torch.float32, device=dev)
assert L.tile16_pack_net(arr, 10, 4, _lib.ptr(p16), _lib.stream_of(p16)) == 0
for M in (1000, 786432):
    x = torch.randn((M, 90), device=dev)
    x[:, :3] *= 2
    out16 = torch.empty((M, 4), device=dev)
    f16 = lambda: L.tile16_mlp_forward(_lib.ptr(p16), _lib.ptr(x), M, 10, 4, _lib.ptr(out16), _lib.stream_of(x))
    with torch.no_grad():
        ref = net(x)
        f16()
        torch.cuda.synchronize()
        print(f"M={M}: max |out16 - out32| = {float((out16 - ref).abs().max()):.3e} (|out| up to {float(ref.abs().max()):.1f})")
        for name, fn in (("32-row, 1 wave/SIMD", lambda: net(x)), ("16-row, 2 waves/SIMD", f16)):
            fn(); fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            print(f"  {name}: {ms:.4f} ms  {M * 2 * 593408 / ms / 1e9:.1f} TFLOP/s")
