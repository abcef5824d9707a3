#!/bin/bash
set -u
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; s=r.get('shares_us_per_frame')
print("%s: %.4g evals/s, %.3f ms per 8 frames, residual %.2f us, solve %.2f us (events)" % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel_us'], s['lidar_solve_kernel_us']))
PY
}
for rep in 1 2; do
  LIVO2_LIDAR_BLOCK_ORDER=1 python bench.py --no-cpu --no-extra > /tmp/on.json 2> /dev/null; show /tmp/on.json "block order ON "
  LIVO2_LIDAR_BLOCK_ORDER=0 python bench.py --no-cpu --no-extra > /tmp/off.json 2> /dev/null; show /tmp/off.json "block order OFF"
done
