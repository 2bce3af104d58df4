#!/bin/bash
# end of round 3: kernel trace of the default bench.py run (headline + extra.configs, incl. the training rows with the grouped
# GEMM launch) and of the headline-only run - the per-kernel averages bench.py's roofline block must agree with
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_final; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 5 > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_bench_with_extras.md
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
grep '^{' $OUT/trace.log > $OUT/bench_under_rocprof.json
rm -rf $OUT/trace; echo "trace ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra > $OUT/trace_c2.log 2>&1 || { echo "c2 trace failed"; exit 1; }
f=$(find $OUT/trace_c2 -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_summary.md
grep '^{' $OUT/trace_c2.log > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/trace_c2; echo "c2 trace ok"
head -30 $OUT/kernel_trace_summary.md
