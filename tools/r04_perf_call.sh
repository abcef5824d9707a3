#!/bin/bash
set -u
O=gpurun_out/r04m; mkdir -p $O
python -m pytest tests/test_lidar_gpu.py tests/test_edge_gpu.py tests/test_bench_workload_gpu.py tests/test_full_size_gpu.py tests/test_configs_gpu.py tests/test_batch_gpu.py tests/test_ref_direct_gpu.py tests/test_host_shim_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest_sel.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_sel.txt
show() { python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d['roofline']; s=r.get('shares_us_per_frame')
print("%s: %.4g evals/s, ms/step %.3f, residual %.2f us (frac %.3f), lidar_solve %.2f, visual/step %.2f" % (sys.argv[2], d['value'], d['ms_per_step'], r['kernel_us'], r['frac'], s['lidar_solve_kernel_us'], r['visual']['kernel_us']))
PY
}
for rep in 1 2; do
  LIVO2_LIDAR_SHARE_RUNS=1 python bench.py --no-cpu --no-extra > $O/b_on$rep.json 2> /dev/null; show $O/b_on$rep.json "share runs ON "
  LIVO2_LIDAR_SHARE_RUNS=0 python bench.py --no-cpu --no-extra > $O/b_off$rep.json 2> /dev/null; show $O/b_off$rep.json "share runs OFF"
done
