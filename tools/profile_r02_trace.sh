#!/bin/bash
# round 2: rocprofv3 kernel trace of the HEADLINE command (bench.py C2, no extras): summary -> gpurun_out/prof_r02h
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02h; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extra > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_summary.md
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log > $OUT/bench_line.json
rm -rf $OUT/trace
cat $OUT/kernel_trace_summary.md | head -8; cat $OUT/bench_line.json | cut -c1-400
