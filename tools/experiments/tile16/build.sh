#!/bin/bash
# builds libtile16.so next to this script (hipcc cross-compiles without a GPU)
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 \
  -I ../../../sw-nerf_amd/csrc -o libtile16.so tile16_kernels.hip
