#!/bin/bash
set -u
O=gpurun_out/r04k; mkdir -p $O
fails=0
for k in $(seq 1 14); do
  b=0xCB; [ $((k % 2)) = 0 ] && b=0xFF
  LIVO2_REDZONE=1 LIVO2_POISON=$b timeout 300 python tests/redzone_frame.py > $O/f_$k.txt 2>&1; rc=$?
  [ $rc != 0 ] && fails=$((fails+1))
  echo "run $k poison $b rc=$rc $(grep -a 'Livo2Error\|Error:' $O/f_$k.txt | tail -1 | cut -c1-300)"
done
echo "failures: $fails of 14"
LIVO2_REDZONE=1 LIVO2_POISON=0xCB timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_redzone_gpu.py -p no:cacheprovider > $O/pytest_rz1_poison.txt 2>&1; echo "suite under redzone+poison rc=$?"; tail -3 $O/pytest_rz1_poison.txt
python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; echo "suite rc=$?"; tail -3 $O/pytest.txt
