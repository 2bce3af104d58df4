#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats, then PMC passes (each its own run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_final; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
echo "trace ok"
run() { name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1 || { echo "$name failed"; return 1; }
  echo "$name ok"; }
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 && \
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run waits SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE && \
run l2 TCC_HIT_sum TCC_MISS_sum
