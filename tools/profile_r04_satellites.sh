#!/bin/bash
# PMC evidence for the satellite kernels, ONE ROW of tools/bench_satellites.py per profiler run (the rows share kernel names
# and grids, so a whole-tool run cannot tell them apart): FETCH_SIZE, WRITE_SIZE and the VALU issue counters in separate passes
# -> gpurun_out/prof_r04_sat/satellites_pmc.md (tools/summarize_satellites_pmc.py)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r04_sat; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
i=0
while IFS= read -r row; do
  i=$((i+1))
  for pass in fetch write valu; do
    case $pass in fetch) ctr="FETCH_SIZE";; write) ctr="WRITE_SIZE";; valu) ctr="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE";; esac
    timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $OUT/row$i/pmc_$pass -- python3 tools/bench_satellites.py --reps 6 --only "$row" > $OUT/row${i}_$pass.log 2>&1 || { echo "row $i $pass failed"; tail -3 $OUT/row${i}_$pass.log; exit 1; }
  done
  echo "$row" > $OUT/row$i/name.txt
  grep -F "$row" $OUT/row${i}_valu.log | grep -v rocprofv3 | head -1 > $OUT/row$i/line.txt
  python3 tools/summarize_pmc_any.py $OUT/row$i "" --skip 3 --json $OUT/row$i/pmc.json > /dev/null
  echo "row $i ok: $row"
done <<'ROWS'
get_rays 4000x4000
pack_ray_batch 16000000 rays -> [N,11]
pack_ray_batch 16000000 rays -> [N,12]
embed 786432 x 3 -> 63
embed 786432 x 1 -> 21
raw2outputs 640000 x 192 (all five
raw2outputs 640000 x 64 (all five
raw2outputs backward 640000 x 192
sample_pdf 640000 x (63 bins -> 128), det
sample_pdf + sort 640000 x (63 bins -> 128 -> 192), det
ROWS
python3 tools/summarize_satellites_pmc.py $OUT > $OUT/satellites_pmc.md
find $OUT -name '*.csv' -delete; find $OUT -type d -empty -delete
cat $OUT/satellites_pmc.md
