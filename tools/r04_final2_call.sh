#!/bin/bash
# one call: A/B tool on the new build, the whole GPU suite, and — only if that is green — the evidence set of the new build
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04v; mkdir -p "$OUT"
TMPDIR=$ROOT/.c4cache timeout 120 python tools/lidar_ab.py --rounds 3 --variants order=1 > "$OUT/ab.txt" 2>&1; tail -2 "$OUT/ab.txt"
timeout 600 python -m pytest tests -m gpu -q > "$OUT/pytest.txt" 2>&1; rc=$?; tail -3 "$OUT/pytest.txt"
[ $rc -eq 0 ] || { echo "GPU suite rc=$rc: no capture"; exit 1; }
SKIP_VIS_PROBE=1 SWEEP_ARGS="8 4" bash tools/capture_profiles.sh r04 > "$OUT/capture.log" 2>&1; echo "capture rc=$?"
tail -3 "$OUT/capture.log"; tail -c 600 gpurun_out/prof_r04/bench.json
