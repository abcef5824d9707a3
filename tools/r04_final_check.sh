#!/bin/bash
set -u
O=gpurun_out/r04_final; mkdir -p $O
python bench.py --no-cpu --no-extra > $O/first_process_bench.json 2> /dev/null; echo "first-process bench rc=$?"
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; echo "suite rc=$?"; tail -3 $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
LIVO2_REDZONE=1 LIVO2_POISON=0xCB timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_redzone_gpu.py -p no:cacheprovider > $O/pytest_rz1_poison.txt 2>&1; echo "suite under redzone+poison rc=$?"; tail -2 $O/pytest_rz1_poison.txt
LIVO2_REDZONE=1 timeout 900 python bench.py --no-cpu --steps 5 > $O/bench_rz1.json 2> $O/bench_rz1.err; echo "bench (all legs) under redzones rc=$?"
