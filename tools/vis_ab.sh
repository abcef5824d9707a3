#!/bin/bash
# A/B of two BUILDS on the resident visual grid: tools/vis_ab.sh <tag> <alt.so>   (ABAB x 2; tools/vis_persist_probe.py at the reference's sub-map size and at C4's, tools/frame_probe.py c1)
set -u
ROOT=$(pwd); TAG=$1; ALT=$2
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=$ROOT/.c4cache; mkdir -p "$TMPDIR"
L=fast-livo2_amd/lib/liblivo2_hip.so
cp $L /tmp/base.so
for rep in 1 2; do
  for v in base "$ALT"; do
    [ "$v" = base ] && cp /tmp/base.so $L || cp "$v" $L
    echo "== $v (rep $rep)" >> "$OUT/ab.txt"
    for M in 350 1000 4000; do timeout 120 python tools/vis_persist_probe.py $M 200 2>&1 | grep "persistent steps" | tail -2 >> "$OUT/ab.txt"; done
    timeout 200 python tools/frame_probe.py c1 64 1 2>&1 | grep "one context" | tail -1 >> "$OUT/ab.txt"
  done
done
cp /tmp/base.so $L
cat "$OUT/ab.txt"
