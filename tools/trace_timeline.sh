#!/bin/bash
# trace_timeline.sh <tag> <python tool + args...>: kernel timeline of the LAST complete step of a training tool under
# rocprofv3 --kernel-trace: start / end / stream of every kernel >= 20 us relative to the step's first forward launch
# (a step = from one coarse forward render_pass launch to the next one) -> gpurun_out/timeline/<tag>.md
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline; mkdir -p $OUT; rm -rf $OUT/trace_$tag; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace_$tag -- python3 "$@" > $OUT/$tag.log 2>&1 || { echo "trace failed"; tail -5 $OUT/$tag.log; exit 1; }
f=$(find $OUT/trace_$tag -name '*kernel_trace.csv' | head -1)
FULL=$OUT/${tag}_full.txt python3 - "$f" "${FWD_PER_STEP:-2}" > $OUT/$tag.md <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
fw = [i for i, r in enumerate(rows) if "render_pass_kernel" in r["Kernel_Name"]]
i0, i1 = fw[-2 * per], fw[-per]
t0 = int(rows[i0]["Start_Timestamp"])
print("| start us | end us | dur us | queue | kernel |")
print("|---|---|---|---|---|")
busy = 0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if e - s >= 20000:
        print(f"| {s / 1e3:.0f} | {e / 1e3:.0f} | {(e - s) / 1e3:.0f} | {r.get('Queue_Id', '?')} | {r['Kernel_Name'].split('(')[0][-60:]} |")
print(f"\nstep = {(int(rows[i1]['Start_Timestamp']) - t0) / 1e3:.0f} us from first forward launch to the next step's")
import os
if os.environ.get("FULL"):                 # every launch of the step with the idle gap in front of it -> <tag>_full.txt
    with open(os.environ["FULL"], "w") as fo:
        prev = None
        for r in rows[i0:i1 + 1]:
            s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
            gap = 0 if prev is None else s - prev
            prev = max(e, prev or 0)
            fo.write(f"{s / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap / 1e3:6.1f}  {r['Kernel_Name'].split('(')[0][-70:]}\n")
# the small launches (< 20 us) of the step, and how long the GPU sat idle between kernels
from collections import defaultdict
small = defaultdict(lambda: [0, 0])
iv = []
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    iv.append((s, e))
    if e - s < 20000:
        k = r["Kernel_Name"].split("(")[0][-50:]
        small[k][0] += 1; small[k][1] += e - s
iv.sort()
busy, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = int(rows[i1]["Start_Timestamp"]) - t0
print(f"GPU busy {busy / 1e3:.0f} us of {span / 1e3:.0f} us: idle between kernels {(span - busy) / 1e3:.0f} us; {sum(v[0] for v in small.values())} launches under 20 us take {sum(v[1] for v in small.values()) / 1e3:.0f} us:")
for k, v in sorted(small.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"  {v[0]:3d} x {k}: {v[1] / 1e3:.0f} us")
PY
grep -i "step" $OUT/$tag.log | grep -v rocprof | tail -3 >> $OUT/$tag.md
rm -rf $OUT/trace_$tag
cat $OUT/$tag.md
