#!/bin/bash
set -u
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; s=r.get('shares_us_per_frame')
print("%s: %.4g evals/s, %.3f ms per 8 frames, residual %.2f us, solve %.2f us (events)" % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel_us'], s['lidar_solve_kernel_us']))
PY
}
python -m pytest tests/test_lidar_gpu.py tests/test_edge_gpu.py tests/test_bench_workload_gpu.py tests/test_full_size_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2 3; do
  LIVO2_KEY_MUL=1 python bench.py --no-cpu --no-extra > /tmp/on.json 2> /dev/null; show /tmp/on.json "key multiply ON "
  LIVO2_KEY_MUL=0 python bench.py --no-cpu --no-extra > /tmp/off.json 2> /dev/null; show /tmp/off.json "key multiply OFF"
done
