#!/bin/bash
# round 4 evidence, one gpurun call: (a) kernel trace of the default bench.py run (headline + extra.configs) and of the
# headline alone, (b) PMC of the headline launches in separate passes (matrix pipe, FETCH_SIZE, WRITE_SIZE) -> per-launch
# traffic of the coarse and the fine pass (profiles/roofline_traffic.json), (c) the HBM/VALU-bound satellites
# (tools/bench_satellites.py): kernel trace, FETCH_SIZE / WRITE_SIZE, VALU issue counters.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
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
pmc() { tag=$1; shift; name=$1; shift; cmd=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${tag}/pmc_$name -- python3 $cmd > $OUT/${tag}_pmc_$name.log 2>&1 || { echo "$tag $name failed"; tail -3 $OUT/${tag}_pmc_$name.log; return 1; }
  echo "$tag $name ok"; }
C2="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
NS="tools/bench_shapes.py north_star"
SAT="tools/bench_satellites.py --reps 5"
pmc c2 mfma "$C2" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc c2 fetch "$C2" FETCH_SIZE && pmc c2 write "$C2" WRITE_SIZE && \
pmc ns mfma "$NS" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc sat fetch "$SAT" FETCH_SIZE && pmc sat write "$SAT" WRITE_SIZE && \
pmc sat valu "$SAT" SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
python3 tools/summarize_pmc_any.py $OUT/c2 render_pass --skip 3 --json $OUT/pmc_c2.json > $OUT/pmc_c2_render_pass.md
python3 tools/summarize_pmc_any.py $OUT/ns render_pass --skip 3 > $OUT/pmc_north_star_render_pass.md
python3 tools/summarize_pmc_any.py $OUT/sat "" --skip 3 --json $OUT/pmc_satellites.json > $OUT/pmc_satellites.md
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_sat -- python3 tools/bench_satellites.py --json $OUT/satellites.json > $OUT/satellites.txt 2>&1 || { echo "satellites trace failed"; exit 1; }
f=$(find $OUT/trace_sat -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_satellites.md
rm -rf $OUT/trace_sat
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ns -- python3 tools/bench_shapes.py north_star > $OUT/trace_ns.log 2>&1 || { echo "north_star trace failed"; exit 1; }
f=$(find $OUT/trace_ns -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_north_star.md
grep -v '^\[' $OUT/trace_ns.log | grep -v amdgpu.ids > $OUT/bench_north_star.txt
rm -rf $OUT/trace_ns
find $OUT -name '*.csv' ! -name kernel_stats.csv -delete; find $OUT -type d -empty -delete
ls $OUT
