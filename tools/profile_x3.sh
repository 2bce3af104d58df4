#!/bin/bash
# round 2, bf16x3 pass: (a) rocprofv3 kernel trace of bench.py --precision bf16x3 (C2 and C5), (b) PMC sets of the C2 run
# (each its own pass): matrix-pipe busy cycles, LDS bank conflicts, L2 fetch.  Summaries -> gpurun_out/prof_x3
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_x3; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for cfg in C2 C5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$cfg -- python3 bench.py --config $cfg --precision bf16x3 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace_$cfg.log 2>&1 || { echo "trace $cfg failed"; tail -5 $OUT/trace_$cfg.log; exit 1; }
  f=$(find $OUT/trace_$cfg -name '*kernel_trace.csv' | head -1)
  python3 tools/summarize_trace.py $f 3 > $OUT/kernel_trace_$cfg.md
  grep '^{' $OUT/trace_$cfg.log > $OUT/bench_line_$cfg.json
  rm -rf $OUT/trace_$cfg
done
pmc() { name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/c2/pmc_$name -- python3 bench.py --precision bf16x3 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; return 1; }
  echo "pmc $name ok"; }
pmc mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS && \
pmc wait SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES && \
pmc fetch FETCH_SIZE
python3 tools/summarize_pmc_any.py $OUT/c2 > $OUT/pmc_c2_x3.md 2>/dev/null
find $OUT -name '*.csv' -delete; find $OUT -type d -empty -delete
cat $OUT/kernel_trace_C2.md | head -12; cut -c1-300 $OUT/bench_line_C2.json; cat $OUT/pmc_c2_x3.md | head -30
