#!/bin/bash
# TCC hit / miss of k_lidar_residual with and without (non-temporal scan loads, per-XCD block order): rocprofv3 counters over tools/lidar_ab.py, one variant per run
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04q; mkdir -p "$OUT"
export TMPDIR=$ROOT/.c4cache
db() { find "$1" -name '*results.db' | head -1; }
for v in "nt=0,xcd=0" "nt=1,xcd=1" "nt=1,xcd=1,order=0"; do
  d=/tmp/pmc_$(echo $v | tr -c 'a-z0-9\n' _); rm -rf $d
  TMPDIR=$ROOT/.c4cache timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $d -o pmc -- python tools/lidar_ab.py --rounds 1 --variants "$v" > /dev/null 2>> "$OUT/pmc.err"
  python tools/kt_summary.py "$(db $d)" "pmc pass (TCC), lidar_ab $v" --split-us 3 | grep -E "^#|k_lidar" >> "$OUT/pmc_l2.txt"
done
cat "$OUT/pmc_l2.txt"
