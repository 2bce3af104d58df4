#!/bin/bash
# rocprofv3 kernel trace of tools/bench_train_dnerf.py; per-kernel table -> gpurun_out/prof_train_dnerf/summary.md
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train_dnerf; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/bench_train_dnerf.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/summary.md <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
g = defaultdict(list)
for r in rows:
    g[(r["Kernel_Name"].split("(")[0][-56:], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("| kernel | grid.x | launches (12 steps: 6 without, 6 with the TV render) | avg us | total ms |")
print("|---|---|---|---|---|")
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) / 1e6 >= 0.5:
        print(f"| {k[0]} | {k[1]} | {len(v)} | {sum(v) / len(v) / 1e3:.1f} | {sum(v) / 1e6:.1f} |")
PY
tail -n 5 $OUT/trace.log; cat $OUT/summary.md; rm -rf $OUT/trace
