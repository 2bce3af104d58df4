#!/bin/bash
# pmc_mfma.sh <tag> <python tool + args...>: matrix-pipe busy fraction and shader clock per kernel of a tool, one rocprofv3 --pmc pass
# (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; dispatches serialised) -> gpurun_out/pmc/<tag>.md
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc; mkdir -p $OUT; rm -rf $OUT/tr_$tag; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/tr_$tag/pmc_mfma -- python3 "$@" > $OUT/$tag.log 2>&1 || { echo "pmc failed"; tail -3 $OUT/$tag.log; exit 1; }
python3 tools/summarize_pmc_any.py $OUT/tr_$tag --skip 2 > $OUT/$tag.md
rm -rf $OUT/tr_$tag
grep -E "render_pass|gemm_tn|narrow5|^\| kernel|^\|---" $OUT/$tag.md
