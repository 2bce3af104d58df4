#!/bin/bash
# rocprofv3 kernel trace + stats of tools/bench_train.py (7 steps: 2 warm-up + 5 timed); per-kernel table -> summary.md
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_train; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/bench_train.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/summary.md <<'PY'
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
print(f"| all kernels | | | | {tot:.1f} |")
PY
tail -3 $OUT/trace.log
cat $OUT/summary.md
rm -rf $OUT/trace
