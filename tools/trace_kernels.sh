#!/bin/bash
# trace_kernels.sh <tag> <python tool + args...>: per-kernel average / min / max duration over a whole run of a tool under
# rocprofv3 --kernel-trace (tools/summarize_trace.py, first 2 launches of each kernel dropped) -> gpurun_out/trace/<tag>.md
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace; mkdir -p $OUT; rm -rf $OUT/trace_$tag; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 "$@" > $OUT/$tag.log 2>&1 || { echo "trace failed"; tail -5 $OUT/$tag.log; exit 1; }
f=$(find $OUT/trace_$tag -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py "$f" 2 | awk 'NR <= 14' > $OUT/$tag.md
rm -rf $OUT/trace_$tag
cat $OUT/$tag.md
