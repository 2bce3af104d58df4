#!/bin/bash
# build.sh <name> [-DTRAIN_EXP_NOBITS | -DTRAIN_EXP_NOSTORE ...]: the library with train_kernels.hip rebuilt under experiment switches
# (the other objects come from sw-nerf_amd/build/, i.e. run __graft_entry__.build() first) -> tools/experiments/train/libswnerf_<name>.so.
# Timing only: such builds compute WRONG gradients.  Use with tools/exp_lib.py or SWNERF_LIB_PATH-style probes.
cd "$(dirname "$0")" && R=../../../sw-nerf_amd && name=$1 && shift && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -w "$@" -c $R/csrc/train_kernels.hip -o train_$name.o && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o libswnerf_$name.so $R/build/render_kernels.o train_$name.o $R/build/misc_kernels.o \
  $R/build/pack_kernels.o $R/build/backward_kernels.o $R/build/generic_kernels.o $R/build/x3_kernels.o
