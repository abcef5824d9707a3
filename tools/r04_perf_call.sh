#!/bin/bash
set -u
O=gpurun_out/r04h; mkdir -p $O
python bench.py --no-cpu --no-extra > $O/bench_head.json 2> $O/bench_head.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench_head.json')); r=d['roofline']
print("value %.4g evals/s, ms/step %.3f, residual %.2f us, shares %s" % (d['value'], d['ms_per_step'], r['kernel_us'], r.get('shares_us_per_frame')))
PY
LIVO2_LIDAR_BLOCK_ORDER=0 python bench.py --no-cpu --no-extra > $O/bench_head_noorder.json 2> /dev/null; python - <<'PY'
import json
d=json.load(open('gpurun_out/r04h/bench_head_noorder.json')); r=d['roofline']
print("block order OFF: value %.4g evals/s, ms/step %.3f, residual %.2f us, shares %s" % (d['value'], d['ms_per_step'], r['kernel_us'], r.get('shares_us_per_frame')))
PY
for g in 250 125; do fast-livo2_amd/lib/exchange_probe $g 200 4000 > $O/exchange_$g.txt 2>&1; cat $O/exchange_$g.txt; done
python -m pytest tests/test_visual_gpu.py tests/test_bench_workload_gpu.py tests/test_redzone_gpu.py tests/test_lidar_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sel.txt
