#!/bin/bash
# run_n5.sh: build the narrow5 experiment variants on the box and time them (tools/probe_n5_exp.py) -> stdout
cd "$(dirname "$0")" || exit 1
bash build.sh n5_ship && bash build.sh n5_nobarrier -DN5_EXP_NOBARRIER && bash build.sh n5_nodma -DN5_EXP_NODMA && bash build.sh n5_nomfma -DN5_EXP_NOMFMA \
  && bash build.sh n5_nodma_nobarrier -DN5_EXP_NODMA -DN5_EXP_NOBARRIER || exit 1
cd ../../.. && timeout -k 10 200 python3 tools/probe_n5_exp.py tools/experiments/gemm/libgemm_n5_ship.so tools/experiments/gemm/libgemm_n5_nobarrier.so \
  tools/experiments/gemm/libgemm_n5_nodma.so tools/experiments/gemm/libgemm_n5_nomfma.so tools/experiments/gemm/libgemm_n5_nodma_nobarrier.so tools/experiments/gemm/libgemm_n5_ship.so
