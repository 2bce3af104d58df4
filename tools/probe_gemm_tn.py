import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT+'/sw-nerf_amd')
import torch
from swnerf import _lib
L=_lib.lib(); dev=torch.device('cuda:0')
M=786432
def run(lda, ldb, No=256, Ni=256, reps=5, label=''):
    A=torch.randn((M,lda),device=dev); B=torch.randn((M,ldb),device=dev)
    C=torch.zeros((No,Ni),device=dev); bias=torch.zeros(No,device=dev)
    st=_lib.stream_of(A)
    f=lambda: _lib.check(L.swnerf_gemm_tn(A.data_ptr(), lda, No, B.data_ptr(), ldb, Ni, M, C.data_ptr(), Ni, bias.data_ptr(), st),'g')
    f(); torch.cuda.synchronize()
    ref=(A[:4096,:No].double().T @ B[:4096,:Ni].double())
    C.zero_(); 
    _lib.check(L.swnerf_gemm_tn(A.data_ptr(), lda, No, B.data_ptr(), ldb, Ni, 4096, C.data_ptr(), Ni, None, st),'g'); torch.cuda.synchronize()
    err=float((C.double()-ref).abs().max()/ref.abs().max())
    t0=time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/reps
    print(f'{label:40s} lda={lda:5d} ldb={ldb:5d} No={No:3d} Ni={Ni:3d}: {dt*1e3:7.3f} ms  {2*M*No*Ni/dt/1e12:6.1f} TFLOP/s  {(M*(No+Ni)*4)/dt/1e12:5.2f} TB/s  relerr {err:.1e}')
    del A,B
run(256,256,label='dense')
run(2432,2432,label='strided as in act/grad')
run(2560,2560,label='stride 2560 (10 KiB rows)')
run(2432,2432,No=128,label='views main')
run(2432,90,Ni=63,label='x part (L0)')
run(4,2432,No=1,label='alpha')
run(4,2432,No=3,Ni=128,label='rgb')
