#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train
mkdir -p $OUT && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_train.py > $OUT/train.log 2>&1; echo "exit=$?" >> $OUT/train.log; tail -3 $OUT/train.log
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_train.py > $OUT/train_traced.log 2>&1
f=$(find $OUT/trace -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -14
