#!/bin/bash
# One parameterised job script for a `gpurun` call (replaces the per-call r04_*_call.sh files): every line of the job is run with the C4 scenario cache on TMPDIR,
# its output collected under gpurun_out/<tag>/, and a one-line verdict per step printed at the end (that tail is what gpurun shows).
#   tools/gpu_call.sh <tag> '<step name>::<command>' ['<step name>::<command>' ...]
# e.g. gpurun --timeout 900 -- 'bash tools/gpu_call.sh r05a "tests::python -m pytest tests/test_lidar_gpu.py -m gpu -x -q" "ab::python tools/lidar_ab.py --rounds 2"'
set -u
ROOT=$(pwd); TAG=$1; shift
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=$ROOT/.c4cache; mkdir -p "$TMPDIR"
SUMMARY=()
for step in "$@"; do
  name=${step%%::*}; cmd=${step#*::}
  t0=$(date +%s)
  ( eval "$cmd" ) > "$OUT/$name.txt" 2>&1; rc=$?
  SUMMARY+=("$name: rc=$rc $(( $(date +%s) - t0 )) s")
  echo "===== $name (rc=$rc)"; tail -${TAIL_LINES:-12} "$OUT/$name.txt"
done
printf '%s\n' "${SUMMARY[@]}"
