#!/bin/bash
# round 3, training step (tools/bench_train.py, 7 steps): kernel trace (the GEMMs of a chunk overlap on two side streams, so the
# per-kernel sum exceeds the step time) and PMC passes - under PMC the dispatches are serialised, which is what the per-kernel
# clock (GRBM_GUI_ACTIVE / duration / 8 XCDs) and matrix-pipe figures need.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_train; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_train.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/train_step_kernels.md <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
g = defaultdict(list)
for r in rows:
    g[(r["Kernel_Name"].split("(")[0][-56:], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
steps = 7
print("| kernel | grid.x | launches per step | avg us | us per step |")
print("|---|---|---|---|---|")
tot = 0.0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / steps / 1e3
    tot += per
    if per >= 5.0:
        print(f"| {k[0]} | {k[1]} | {len(v) / steps:.1f} | {sum(v) / len(v) / 1e3:.1f} | {per:.1f} |")
print(f"| all kernels (overlapping launches counted in full) | | | | {tot:.1f} |")
PY
grep "training step\|peak memory" $OUT/trace.log > $OUT/bench_train_under_trace.txt
rm -rf $OUT/trace
for set in "mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
  set -- $set; name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/tr/pmc_$name -- python3 tools/bench_train.py > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 $OUT/pmc_$name.log; exit 1; }
  echo "pmc $name ok"
done
python3 tools/summarize_pmc_any.py $OUT/tr --skip 2 > $OUT/pmc_all.md
grep -E "render_pass|gemm_tn|narrow5|feature_finish|^\| kernel|^\|---" $OUT/pmc_all.md > $OUT/pmc_train_step.md
find $OUT -name '*.csv' -delete; find $OUT -type d -empty -delete
cat $OUT/train_step_kernels.md | head -20; cat $OUT/pmc_train_step.md
