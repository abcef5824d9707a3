#!/bin/bash
# ABAB of BUILDS of the library under any command: tools/cmd_ab.sh <tag> "<command>" <alt1.so> [alt2.so ...]   (the command's stdout is filtered by $AB_GREP if set)
set -u
ROOT=$(pwd); TAG=$1; CMD=$2; shift 2
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
L=fast-livo2_amd/lib/liblivo2_hip.so
cp $L /tmp/base.so
for rep in 1 2 3; do
  for v in base "$@"; do
    [ "$v" = base ] && cp /tmp/base.so $L || cp "$v" $L
    echo "== $v (rep $rep)" | tee -a "$OUT/ab.txt"
    timeout 300 bash -c "$CMD" 2>&1 | grep -E "${AB_GREP:-.}" | tee -a "$OUT/ab.txt"
  done
done
cp /tmp/base.so $L
