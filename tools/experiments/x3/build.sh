#!/bin/bash
# Timing variants of the bf16x3 pass (never shipped; outputs of the NOBARRIER / NOWAIT builds are garbage by construction):
#   build.sh <name> [-DX3_EXP_NOBARRIER] [-DX3_EXP_NOWAIT] ...   ->  libswnerf_<name>.so
cd "$(dirname "$0")" && C=../../../sw-nerf_amd/csrc && name=$1 && shift && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "$@" \
  -o libswnerf_$name.so $C/render_kernels.hip $C/train_kernels.hip $C/misc_kernels.hip $C/pack_kernels.hip $C/backward_kernels.hip $C/generic_kernels.hip $C/x3_kernels.hip
