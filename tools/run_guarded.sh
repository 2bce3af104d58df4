#!/bin/bash
# run_guarded.sh <log> <command ...> - run a GPU tool with its output in <log>, show the tail, and FAIL (exit 97) when the run
# left a GPU fault line in it even if the command's own exit code got lost on the way (round 3: a probe that died with
# "Memory access fault by GPU node-2" sat in a `cmd > log; cat log` chain whose status was cat's).  Use it for every probe /
# soak / bench step of a gpurun call and join the steps with `&&`.
log=$1; shift
mkdir -p "$(dirname "$log")"
"$@" > "$log" 2>&1
rc=$?
tail -n "${GUARD_TAIL:-12}" "$log"
if grep -q -e "Memory access fault" -e "HSA_STATUS_ERROR" -e "Segmentation fault" "$log"; then
  echo "[run_guarded] GPU fault line in $log (command exit code $rc)" >&2
  exit 97
fi
exit $rc
