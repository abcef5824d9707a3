#!/bin/bash
# A/B of two builds of the library on one box: r04_lib_ab.sh <alt .so> [pytest selection...]   (tools/lidar_ab.py per build, ABAB; then the GPU tests named with the alt build)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04s; mkdir -p "$OUT"
ALT=$1; shift
L=fast-livo2_amd/lib/liblivo2_hip.so
cp $L /tmp/base.so; cp $ALT /tmp/alt.so
for rep in 1 2 3; do
  for v in base alt; do
    cp /tmp/$v.so $L
    echo "== $v (rep $rep)" >> "$OUT/ab.txt"
    TMPDIR=$ROOT/.c4cache timeout 120 python tools/lidar_ab.py --rounds 1 --variants order=1 >> "$OUT/ab.txt" 2>&1
  done
done
grep -E "^==|^order" "$OUT/ab.txt"
if [ $# -gt 0 ]; then
  cp /tmp/alt.so $L
  timeout 600 python -m pytest "$@" -m gpu -x -q 2>&1 | tail -4
fi
cp /tmp/base.so $L
