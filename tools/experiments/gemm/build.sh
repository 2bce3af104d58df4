#!/bin/bash
# build.sh <name> [-DGEMM_EXP_...]: a small library with the weight-gradient kernels alone (backward_kernels.hip + the error buffer
# of misc_kernels.hip) for timing experiments on the 256 x 256 GEMM group (tools/probe_gemm_exp.py).  Not the shipped build.
cd "$(dirname "$0")" && C=../../../sw-nerf_amd/csrc && name=$1 && shift && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -w "$@" \
  -o libgemm_$name.so $C/backward_kernels.hip $C/misc_kernels.hip
