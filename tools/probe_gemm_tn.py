#!/usr/bin/env python3
"""swnerf_gemm_tn timing at the shapes of a training step (event-timed, per launch).
usage: probe_gemm_tn.py [M ...]   default M = 786432 (C2 fine pass rows)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/sw-nerf_amd')
import torch
from swnerf import _lib
if os.environ.get("SWNERF_LIB"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SWNERF_LIB"])
L = _lib.lib()
dev = torch.device('cuda:0')


def run(M, lda, ldb, No=256, Ni=256, reps=10, label=''):
    A = torch.randn((M, lda), device=dev)
    B = torch.randn((M, ldb), device=dev)
    C = torch.zeros((No, Ni), device=dev)
    bias = torch.zeros(No, device=dev)
    st = _lib.stream_of(A)
    f = lambda m=M: _lib.check(L.swnerf_gemm_tn(A.data_ptr(), lda, No, B.data_ptr(), ldb, Ni, m, C.data_ptr(), Ni, bias.data_ptr(), st), 'g')
    f()
    torch.cuda.synchronize()
    mc = min(M, 8192)
    ref = (A[:mc, :No].double().T @ B[:mc, :Ni].double())
    C.zero_()
    f(mc)
    torch.cuda.synchronize()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    dt = e0.elapsed_time(e1) / reps * 1e-3
    print(f'| {label:28s} | M={M:7d} lda={lda:5d} ldb={ldb:5d} No={No:3d} Ni={Ni:3d} | {dt*1e3:7.3f} ms | {2*M*No*Ni/dt/1e12:6.1f} TFLOP/s | '
          f'{(M*(No+Ni)*4)/dt/1e12:5.2f} TB/s | relerr {err:.1e} |')
    del A, B


for M in ([int(a) for a in sys.argv[1:]] or [786432]):
    run(M, 256, 256, label='dense')
    run(M, 2432, 2432, label='windows of act/grad')
    run(M, 2432, 2432, No=128, label='views main')
    run(M, 2432, 90, Ni=63, label='x part (L0)')
    run(M, 4, 2432, No=1, label='alpha')
    run(M, 4, 2432, No=3, Ni=128, label='rgb')
