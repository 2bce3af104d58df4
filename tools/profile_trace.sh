#!/bin/bash
# kernel trace + stats of the bench command (no counters in this pass)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r01
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_under_trace.log 2>&1
echo "trace exit=$?" >> $OUT/bench_under_trace.log
find $OUT/trace -name "*stats*" | head; find $OUT/trace -name "*kernel_stats*" -exec head -20 {} \;
