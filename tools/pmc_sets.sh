#!/bin/bash
# !! the TCP_UTCL1_* / TCP_* / TA_* sets below made rocprofv3 abort in hipStreamCreate on this image (signal 6) and each run then sat out its timeout: 10 GPU-minutes
#    for nothing.  Only the first two sets are known good; try the others one at a time with `timeout 30`.
# counter passes over tools/lidar_ab.py (torch-free C4 frame updates): one rocprofv3 --pmc run per set, summary per kernel -> gpurun_out/pmc_sets/pmc_sets.txt
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_sets; mkdir -p "$OUT"; : > "$OUT/pmc_sets.txt"
db() { find "$1" -name '*results.db' | head -1; }
k=0
while read -r set; do
  [ -z "$set" ] && continue
  k=$((k+1)); d=/tmp/pmcset_$k; rm -rf $d
  TMPDIR=$ROOT/.c4cache timeout 40 rocprofv3 --pmc $set --kernel-trace -d $d -o pmc -- python tools/lidar_ab.py --rounds 1 --variants order=1 > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(db $d)" "pmc pass ($set), lidar_ab" --split-us 3 | grep -E "^#|k_lidar_residual|k_visual_update" >> "$OUT/pmc_sets.txt"
done <<'SETS'
SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES
SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD
TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS TCP_UTCL1_THRASHING_STALL
TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TOTAL_ACCESSES TCP_TCR_TCP_STALL_CYCLES
TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_READ_WAVEFRONTS TD_TD_BUSY TD_TC_STALL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL
SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH
SETS
grep -v "^# .*kernel trace" "$OUT/pmc_sets.txt" | cut -c1-200
tail -5 "$OUT/pmc.err"
