#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/capture_profiles.sh r01'): the bench line, the rocprofv3 kernel trace of
# the same command, and the PMC passes (each in its own run, counters only with --kernel-trace) -> gpurun_out/prof_<tag>/ .
# Copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/kt; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python bench.py --no-cpu > /dev/null 2> "$OUT/kt.err"
python tools/kt_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu (steps=200 warmup=20 batch=16)" > "$OUT/kernel_trace_stats.txt"
find /tmp/kt -name '*kernel_stats*' -exec cp {} "$OUT/" \; 2>/dev/null
: > "$OUT/pmc.txt"
k=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  k=$((k+1)); rm -rf /tmp/pmc$k
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc$k -o pmc -- python bench.py --no-cpu --no-extra --steps 20 --warmup 2 > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(find /tmp/pmc$k -name '*results.db' | head -1)" "pmc pass $k ($set), bench.py --no-cpu --no-extra --steps 20 --warmup 2" | grep -E "^#|k_lidar|k_visual" >> "$OUT/pmc.txt"
done
make -C fast-livo2_amd/csrc prof > /dev/null 2>&1
timeout 200 python tools/phase_prof.py voxelgrid > "$OUT/phase_stamps.txt" 2>&1
ls -la "$OUT"
