set -u
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06dec
timeout 900 python -m pytest tests/test_visual_gpu.py tests/test_visual_inverse_gpu.py tests/test_c5_gpu.py tests/test_bench_workload_gpu.py -x -q > gpurun_out/r06dec/pytest.txt 2>&1
tail -3 gpurun_out/r06dec/pytest.txt
export TMPDIR=$GRAFT_REPO_ROOT/.c4cache; mkdir -p $TMPDIR
L=fast-livo2_amd/lib/liblivo2_hip.so
cp $L /tmp/new.so
for rep in 1 2; do for v in prev new; do
  [ $v = new ] && cp /tmp/new.so $L || cp fast-livo2_amd/lib/liblivo2_hip_prev2.so $L
  echo "== $v (rep $rep)" >> gpurun_out/r06dec/ab.txt
  for M in 350 4000; do LIVO2_PROBE_ONLY=persistent timeout 120 python tools/vis_persist_probe.py $M 300 2>&1 | grep "steps" | tail -2 >> gpurun_out/r06dec/ab.txt; done
done; done
cp /tmp/new.so $L
cat gpurun_out/r06dec/ab.txt
timeout 200 python tools/vis_persist_probe.py 4000 100 > gpurun_out/r06dec/probe.txt 2>&1
grep -A14 "^# block" gpurun_out/r06dec/probe.txt | head -16
