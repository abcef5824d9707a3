#!/bin/bash
# every dispatch and copy of the last frames of host/live_chain (lean) at one size, from a rocprofv3 kernel + memory-copy trace: tools/live_trace.sh <tag> [size=avia] [N=160]
set -u
ROOT=$(pwd); TAG=$1; SIZE=${2:-avia}; N=${3:-160}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
D=/tmp/livo2_live_${SIZE}_v3
python - <<PY
import os, sys
sys.path.insert(0, "$ROOT")
from scenarios import live_inputs
if not os.path.exists(os.path.join("$D", "chain_cfg.bin")):
    live_inputs.write_live_dir("$D", live_inputs.make_live(**live_inputs.SIZES["$SIZE"]))
PY
EXE=$ROOT/fast-livo2_amd/lib/live_chain
LIVO2_SHIM_PROF=1 $EXE $D lean > "$OUT/plain.txt" 2> "$OUT/plain_err.txt"; tail -2 "$OUT/plain.txt"
cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace -d "$OUT/trace" -o live -- $EXE $D lean > "$OUT/traced.txt" 2>&1
DB=$(find "$OUT/trace" -name "*.db" | head -1)
python $ROOT/tools/kt_frame.py "$DB" $N > "$OUT/last_dispatches.txt"
tail -$N "$OUT/last_dispatches.txt" | cut -c1-150
rm -rf "$OUT/trace"
