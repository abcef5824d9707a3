#!/bin/bash
# static look at one kernel of the library (no GPU needed): isa_stats.sh <mangled-name-prefix> [extra hipcc flags...]  ->  /tmp/isa/<prefix>.s + spill / opcode summary
set -e
K=$1; shift
mkdir -p /tmp/isa; cd /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I/root/repo/include --cuda-device-only -c /root/repo/fast-livo2_amd/csrc/livo2_api.hip -o dev.o "$@" 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=dev.o --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=dev.co
/opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn dev.co 2>/dev/null | awk -v k="<$K" 'index($0,k) && /^[0-9a-f]+ </{p=1;next} /^[0-9a-f]+ </{if(p)exit} p{print}' > $K.s
echo "instructions: $(wc -l < $K.s)  v_readlane $(grep -c v_readlane $K.s)  v_writelane $(grep -c v_writelane $K.s)  s_nop $(grep -c s_nop $K.s)  scratch $(grep -c scratch_ $K.s)"
awk '{n++; if($1=="v_readlane_b32")r[int(n/250)]++; if($1=="v_writelane_b32")w[int(n/250)]++; if($1=="s_barrier")b[int(n/250)]++; if($1 ~ /^v_.*f64/)f[int(n/250)]++} END{for(i=0;i<=n/250;i++) printf "%5d: readlane %3d writelane %3d barrier %2d f64 %3d\n", i*250, r[i], w[i], b[i], f[i]}' $K.s
