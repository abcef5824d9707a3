#!/bin/bash
# builds variant libraries for an A/B on the GPU box: tools/build_variants.sh name1:"-DFLAG ..." name2:"..."  ->  fast-livo2_amd/lib/liblivo2_hip_<name>.so (and the default library)
cd /root/repo/fast-livo2_amd/csrc || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -w"
/opt/rocm/bin/hipcc $F -o ../lib/liblivo2_hip.so livo2_api.hip &
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  /opt/rocm/bin/hipcc $F $flags -o ../lib/liblivo2_hip_$name.so livo2_api.hip &
done
wait
ls -la ../lib/*.so
