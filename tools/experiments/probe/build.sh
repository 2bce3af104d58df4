#!/bin/bash
# the library with -DSW_PROBE (csrc/render_pass.h): shader-clock stamps around the parts of a tile.  Not the shipped build.
cd "$(dirname "$0")" && C=../../../sw-nerf_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DSW_PROBE \
  -o libswnerf_probe.so $C/render_kernels.hip $C/train_kernels.hip $C/misc_kernels.hip $C/pack_kernels.hip $C/backward_kernels.hip $C/generic_kernels.hip $C/x3_kernels.hip
