import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT+'/sw-nerf_amd')
import torch
from swnerf import _lib
L=_lib.lib(); dev=torch.device('cuda:0')
M=786432; lda=ldb=2432; No=Ni=256
A=torch.randn((M,lda),device=dev); B=torch.randn((M,ldb),device=dev); C=torch.zeros((No,Ni),device=dev); bias=torch.zeros(No,device=dev)
st=_lib.stream_of(A)
for _ in range(6):
    _lib.check(L.swnerf_gemm_tn(A.data_ptr(), lda, No, B.data_ptr(), ldb, Ni, M, C.data_ptr(), Ni, bias.data_ptr(), st),'g')
torch.cuda.synchronize()
