#!/bin/bash
# round 3: kernel trace of the fused use_viewdirs=False training step (tools/probe_noview_train.py fused: 2 + 7 steps)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r03_noview_train; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/probe_noview_train.py fused > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/noview_train_step_kernels.md <<'PY'
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = defaultdict(list)
# the last 7 steps: everything after the 2 warm-up steps = the last 7/9 of the render_pass launches
first = [i for i, r in enumerate(rows) if "render_pass_kernel" in r["Kernel_Name"]]
cut = first[len(first) * 2 // 9]
for r in rows[cut:]:
    g[(r["Kernel_Name"].split("(")[0][-60:], r["Grid_Size_X"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
steps = 7
print("| kernel | grid.x | launches per step | avg us | us per step |")
print("|---|---|---|---|---|")
tot = 0.0
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1])):
    per = sum(v) / steps / 1e3
    tot += per
    if per >= 5.0:
        print(f"| {k[0]} | {k[1]} | {len(v) / steps:.1f} | {sum(v) / len(v) / 1e3:.1f} | {per:.1f} |")
print(f"| all kernels (the GEMMs of a chunk overlap on two side streams: counted in full) | | | | {tot:.1f} |")
PY
grep "fused\|peak memory" $OUT/trace.log > $OUT/step_under_trace.txt
rm -rf $OUT/trace
cat $OUT/noview_train_step_kernels.md $OUT/step_under_trace.txt
