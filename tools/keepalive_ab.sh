#!/bin/bash
# GPU box: tools/lidar_ab.py alone, beside a keep-alive process (tools/keepalive_probe.hip), alone again — is the slow mode of k_lidar_solve a power state that HBM traffic prevents?
export TMPDIR=$PWD/.c4cache
for mode in alone keepalive alone keepalive; do
  if [ $mode = keepalive ]; then ./fast-livo2_amd/lib/keepalive_probe 6 ${1:-4} ${2:-0} > /tmp/ka.txt 2>&1 & KA=$!; sleep 1; fi
  echo "== $mode"; timeout 100 python tools/lidar_ab.py --rounds 1 --variants order=1 2>&1 | grep "^order"
  if [ $mode = keepalive ]; then wait $KA; cat /tmp/ka.txt; fi
done
