#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icache; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d $OUT/pmc_ic -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/ic.log 2>&1 || { echo failed; tail -5 $OUT/ic.log; }
timeout -k 10 200 rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_if -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > $OUT/if.log 2>&1 || { echo failed2; tail -5 $OUT/if.log; }
python3 tools/summarize_pmc_any.py $OUT render_pass
find $OUT -name '*.csv' -delete
