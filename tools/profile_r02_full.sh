#!/bin/bash
# round 2 evidence: (a) kernel trace of bench.py incl. the extra.configs block, (b) PMC of the headline (C2) launches,
# (c) PMC of the D-NeRF pass (bench.py --config C5), (d) kernel trace + PMC of the training step.  Each PMC set is its own run.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 tools/summarize_trace.py $f 0 > $OUT/kernel_trace_all.md
grep '^{' $OUT/trace.log > $OUT/bench_under_rocprof.json
rm -rf $OUT/trace; echo "trace ok"
pmc() { tag=$1; shift; name=$1; shift; cmd=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/${tag}/pmc_$name -- python3 $cmd > $OUT/${tag}_pmc_$name.log 2>&1 || { echo "$tag $name failed"; tail -3 $OUT/${tag}_pmc_$name.log; return 1; }
  echo "$tag $name ok"; }
C2="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
C5="bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra"
TR="tools/bench_train.py"
pmc c2 mfma "$C2" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc c2 lds "$C2" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS && \
pmc c2 fetch "$C2" FETCH_SIZE && pmc c2 write "$C2" WRITE_SIZE && \
pmc c5 mfma "$C5" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc c5 fetch "$C5" FETCH_SIZE && pmc c5 write "$C5" WRITE_SIZE && \
pmc train mfma "$TR" SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE && \
pmc train fetch "$TR" FETCH_SIZE && pmc train write "$TR" WRITE_SIZE
for t in c2 c5 train; do python3 tools/summarize_pmc_any.py $OUT/$t > $OUT/pmc_$t.md 2>/dev/null; done
find $OUT -name '*.csv' -delete; find $OUT -type d -empty -delete
ls $OUT
