#!/bin/bash
# kernel timeline of ONE training step (tools/bench_train.py under rocprofv3 --kernel-trace): start / end of every kernel >= 20 us
# relative to the step's first kernel, with the stream (queue) it ran on - to see what overlaps what in the backward.
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 tools/bench_train.py > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" > $OUT/train_step_timeline.md <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one step = from a coarse forward TRAIN launch to the next; take the last complete one
fw = [i for i, r in enumerate(rows) if "render_pass_kernel" in r["Kernel_Name"]]
# forward launches come in pairs (coarse, fine) per step
i0, i1 = fw[-4], fw[-2]
t0 = int(rows[i0]["Start_Timestamp"])
print("| start us | end us | dur us | queue | kernel |")
print("|---|---|---|---|---|")
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if e - s >= 20000:
        print(f"| {s / 1e3:.0f} | {e / 1e3:.0f} | {(e - s) / 1e3:.0f} | {r.get('Queue_Id', '?')} | {r['Kernel_Name'].split('(')[0][-60:]} |")
PY
grep "training step" $OUT/trace.log >> $OUT/train_step_timeline.md
rm -rf $OUT/trace
cat $OUT/train_step_timeline.md
