#!/bin/bash
# A/B of BUILDS of the library on one box (ABAB...): tools/lib_ab.sh <tag> <variant args for lidar_ab.py or ""> <alt1.so> [alt2.so ...] [-- pytest selection...]
# Every build is measured with tools/lidar_ab.py (torch-free, ~3 s), `reps` times round-robin; then the named GPU tests run with EACH alternative build.
set -u
ROOT=$(pwd); TAG=$1; VARIANTS=${2:-order=1}; shift 2
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=$ROOT/.c4cache; mkdir -p "$TMPDIR"
L=fast-livo2_amd/lib/liblivo2_hip.so
ALTS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ALTS+=("$1"); shift; done
[ $# -gt 0 ] && shift
cp $L /tmp/base.so
for rep in 1 2 3; do
  for v in base "${ALTS[@]}"; do
    [ "$v" = base ] && cp /tmp/base.so $L || cp "$v" $L
    echo "== $v (rep $rep)" >> "$OUT/ab.txt"
    timeout 120 python tools/lidar_ab.py --rounds 1 --variants $VARIANTS >> "$OUT/ab.txt" 2>&1
  done
done
grep -E "^==|^[a-z_=0-9,]+ +k_lidar" "$OUT/ab.txt"
if [ $# -gt 0 ]; then
  for v in "${ALTS[@]}"; do
    cp "$v" $L
    echo "== tests with $v"; timeout 600 python -m pytest "$@" -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
  done
fi
cp /tmp/base.so $L
