#!/bin/bash
# Run ON THE GPU BOX from the repo root (gpurun -- 'bash tools/capture_profiles.sh r03'): the bench line, the rocprofv3 kernel trace of the headline-only
# command (one workload per trace: C4 frame updates), and the PMC passes (each in its own run, counters only with --kernel-trace) -> gpurun_out/prof_<tag>/ .
# Copy what should be judged into profiles/ afterwards.
set -u
TAG=${1:-r03}
ROOT=$(pwd); export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
HEAD="python bench.py --no-cpu --no-extra --steps 10 --warmup 2"
timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $HEAD > /dev/null 2> "$OUT/kt.err"
python tools/kt_summary.py "$(find /tmp/kt -name '*results.db' | head -1)" "rocprofv3 --kernel-trace --stats -- $HEAD (C4 frame updates only)" --split-us 3 > "$OUT/kernel_trace_stats_c4.txt"
find /tmp/kt -name '*kernel_stats*' -exec cp {} "$OUT/" \; 2>/dev/null
: > "$OUT/pmc_c4.txt"
k=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  k=$((k+1)); rm -rf /tmp/pmc$k
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc$k -o pmc -- $HEAD > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(find /tmp/pmc$k -name '*results.db' | head -1)" "pmc pass $k ($set), $HEAD" --split-us 3 | grep -E "^#|k_lidar|k_visual" >> "$OUT/pmc_c4.txt"
done
# the lockstep / batched legs: one trace of the full bench (mixed workloads, kernels of the batched paths have their own names)
rm -rf /tmp/kt2; timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt2 -o kt -- python bench.py --no-cpu > /dev/null 2> "$OUT/kt2.err"
python tools/kt_summary.py "$(find /tmp/kt2 -name '*results.db' | head -1)" "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu (all legs)" --split-us 3 > "$OUT/kernel_trace_stats_all.txt"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  k=$((k+1)); rm -rf /tmp/pmc$k
  timeout 900 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc$k -o pmc -- python bench.py --no-cpu > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(find /tmp/pmc$k -name '*results.db' | head -1)" "pmc pass $k ($set), python bench.py --no-cpu (all legs)" --split-us 3 | grep -E "^#|_batch" >> "$OUT/pmc_batched.txt"
done
ls -la "$OUT"
# round 3: the persistent visual update against the launch-per-step sequence, and the parity sweep with this library
timeout 300 python tools/vis_persist_probe.py > "$OUT/vis_persist_probe.txt" 2> "$OUT/vis_persist_probe.err"
timeout 900 python tests/sweeps/parity_sweep.py 12 8 > "$OUT/parity_sweep.txt" 2> "$OUT/parity_sweep.err"
ls -la "$OUT"
