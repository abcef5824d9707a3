#!/bin/bash
# ONE consistent evidence set from the FINAL binary, in one call ON THE GPU BOX from the repo root (gpurun -- 'bash tools/capture_profiles.sh r04'):
#   bench line (default command) + full report, rocprofv3 kernel trace of the headline-only command, the PMC passes of the headline (each in its own run, counters
#   only with --kernel-trace), and a traffic record for C4, lockstep, batched AND out-of-cache (tools/traffic.py: every record carries the csrc hash of this tree,
#   bench.py refuses records of any other build).  Everything lands in gpurun_out/prof_<tag>/ ; copy what should be judged into profiles/ afterwards
#   (tools/capture_profiles.sh does not write under profiles/ itself: gpurun only merges gpurun_out/ back).
set -u
TAG=${1:-r04}
ROOT=$(pwd); export TMPDIR=/tmp
cp .c4cache/livo2_c4_*.pkl .c4cache/livo2_c5_*.pkl /tmp/ 2>/dev/null          # the headline frame, if a generated copy travelled with the tree (bench.c4_frame caches it under $TMPDIR; ~40 s of numpy otherwise)
OUT=$ROOT/gpurun_out/prof_$TAG; mkdir -p "$OUT"
python tools/traffic.py sha > "$OUT/build_sha.txt"
HEAD="python bench.py --no-cpu --no-extra --steps 10 --warmup 2"
db() { find "$1" -name '*results.db' | head -1; }
pmc3() {   # pmc3 <workload> <command...>: FETCH / WRITE / TCC passes of one command -> traffic record
  local wl=$1; shift
  local k=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    k=$((k+1)); rm -rf /tmp/pmc_${wl}_$k
    timeout 900 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_${wl}_$k -o pmc -- "$@" > /dev/null 2>> "$OUT/pmc.err"
    python tools/kt_summary.py "$(db /tmp/pmc_${wl}_$k)" "pmc pass ($set), $*" --split-us 3 | grep -E "^#|k_lidar|k_visual" >> "$OUT/pmc_${wl}.txt"
  done
  python tools/traffic.py make-from-bench "$TAG" "$wl" "$(db /tmp/pmc_${wl}_1)" "$(db /tmp/pmc_${wl}_2)" "$(db /tmp/pmc_${wl}_3)" gpurun_out/bench_full.json "$*" >> "$OUT/traffic.log" 2>&1
  cp gpurun_out/${TAG}_traffic_${wl}.json "$OUT/" 2>/dev/null
}
# 0. the GPU suite with this binary (first: everything below is evidence about a library that must be correct)
if [ -z "${SKIP_SUITE:-}" ]; then
  timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_gpu.txt" 2>&1 || { echo "GPU suite failed: no capture"; tail -30 "$OUT/pytest_gpu.txt"; exit 1; }
fi
# 1. the kernel trace of the headline-only command (one workload per trace: C4 frame updates)
rm -rf /tmp/kt; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- $HEAD > /dev/null 2> "$OUT/kt.err"
python tools/kt_summary.py "$(db /tmp/kt)" "rocprofv3 --kernel-trace --stats -- $HEAD (C4 frame updates only)" --split-us 3 > "$OUT/kernel_trace_stats_c4.txt"
find /tmp/kt -name '*kernel_stats*' -exec cp {} "$OUT/" \; 2>/dev/null
# 2. wave-level counters of the headline (QUICK=1 skips them, and runs the default bench instead of --full)
: > "$OUT/pmc_c4_waves.txt"; k=0
[ -n "${QUICK:-}" ] || for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  k=$((k+1)); rm -rf /tmp/pmcw$k
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmcw$k -o pmc -- $HEAD > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(db /tmp/pmcw$k)" "pmc pass ($set), $HEAD" --split-us 3 | grep -E "^#|k_lidar|k_visual" >> "$OUT/pmc_c4_waves.txt"
done
# 3. traffic records: one workload per command
pmc3 c4 $HEAD
pmc3 c4_lockstep python bench.py --no-cpu --c5-frames 0 --legs lockstep --steps 5 --warmup 1
pmc3 batched python bench.py --no-cpu --c5-frames 0 --legs batched --steps 5 --warmup 1
pmc3 out_of_cache python bench.py --no-cpu --c5-frames 0 --legs ooc --steps 5 --warmup 1
# 4. the records are in gpurun_out/<tag>_traffic_*.json; make them visible to the bench run below the way the committed ones will be (profiles/ on this box only)
cp gpurun_out/${TAG}_traffic_*.json profiles/ 2>/dev/null
# 5. the bench line (default command + --full: every informational leg in the full report; the stdout line is the default command's), LAST (its roofline.traffic now cites the records above), then the probes
BENCH_ARGS="--full"; [ -n "${QUICK:-}" ] && BENCH_ARGS=""
timeout 1200 python bench.py $BENCH_ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"; cp gpurun_out/bench_full.json "$OUT/bench_full.json"
[ -n "${SKIP_VIS_PROBE:-}" ] || timeout 300 python tools/vis_persist_probe.py > "$OUT/vis_persist_probe.txt" 2> "$OUT/vis_persist_probe.err"
# 6. the C1-shaped frame through the whole-frame API (the reference's operating point): rates, and the kernel trace of its launch chain
timeout 200 python tools/frame_probe.py c1 64 3 > "$OUT/frame_api_probe.txt" 2> "$OUT/frame_api_probe.err"
rm -rf /tmp/ktf; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ktf -o kt -- python tools/frame_probe.py c1 64 1 > /dev/null 2>> "$OUT/kt.err"
python tools/kt_summary.py "$(db /tmp/ktf)" "rocprofv3 --kernel-trace --stats -- python tools/frame_probe.py c1 64 1 (C1-shaped frames, livo2_frame_update_async / _fetch)" --split-us 3 > "$OUT/kernel_trace_stats_c1_frame.txt"
timeout 900 python tests/sweeps/parity_sweep.py ${SWEEP_ARGS:-12 8} > "$OUT/parity_sweep.txt" 2> "$OUT/parity_sweep.err"
ls -la "$OUT"
