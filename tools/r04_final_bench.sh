#!/bin/bash
set -u
O=gpurun_out/prof_r04; mkdir -p $O
cp profiles/r04_traffic_*.json /dev/null 2>&1
timeout 1200 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cp gpurun_out/bench_full.json $O/bench_full.json
LIVO2_SHIM_PROF=1 fast-livo2_amd/lib/live_chain /tmp/livo2_live_c4_v3 lean 2>&1 | grep -a "^frame\|live_chain\|StateEst" | cut -c1-200
