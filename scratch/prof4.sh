#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_gemm; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/a -- python3 scratch/gemm_probe2.py > $OUT/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD --kernel-trace --output-format csv -d $OUT/b -- python3 scratch/gemm_probe2.py > $OUT/b.log 2>&1
for d in a b; do f=$(ls $OUT/$d/*/*counter_collection.csv); grep gemm_tn $f | awk -F, '{n[$16]+=$17; c[$16]++} END{for(k in n) print k, n[k]/c[k]*1.0}' ; done
