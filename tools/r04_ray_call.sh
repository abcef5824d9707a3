#!/bin/bash
set -u
O=gpurun_out/r04q; mkdir -p $O
python -m pytest tests/test_raycast_gpu.py tests/test_select_gpu.py tests/test_retrieve_chain_gpu.py tests/test_host_shim_gpu.py tests/test_live_chain_gpu.py -m gpu -q -x -p no:cacheprovider > $O/pytest.txt 2>&1; echo "rc=$?"; tail -25 $O/pytest.txt | cut -c1-250
