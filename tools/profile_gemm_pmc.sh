#!/bin/bash
# PMC counters of the weight-gradient GEMM (tools/probe_gemm_tn.py), each set in its own run
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gemm; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 tools/probe_gemm_tn.py 786432 > $OUT/pmc_$name.log 2>&1 || { echo "$name failed"; return 1; }
  echo "$name ok"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES && \
run waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && \
run insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES
python3 tools/summarize_pmc_any.py $OUT gemm_tn > $OUT/summary.md
find $OUT -name '*.csv' -size +2M -delete
cat $OUT/summary.md
