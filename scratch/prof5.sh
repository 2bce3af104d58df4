#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train2; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_train.py > $OUT/train_traced.log 2>&1
tail -2 $OUT/train_traced.log | cut -c1-200
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-110 $f | head -8
