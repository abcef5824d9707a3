#!/bin/bash
set -u
O=gpurun_out/r04g; mkdir -p $O
S=fast-livo2_amd/lib/fence_selftest
for m in 2 3; do LIVO2_FENCE_GRAN=recommended LIVO2_REDZONE=$m timeout 120 $S > $O/selftest_rec_$m.txt 2>&1; echo "selftest recommended-granule mode $m rc=$?"; grep -a -v "^  0x\|^  freed" $O/selftest_rec_$m.txt | tail -2; done
for m in 2 3; do LIVO2_FENCE_GRAN=recommended AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3 LIVO2_REDZONE=$m timeout 120 $S > $O/selftest_rec_ser_$m.txt 2>&1; echo "selftest recommended + serialised mode $m rc=$?"; grep -a -v "^  0x\|^  freed" $O/selftest_rec_ser_$m.txt | tail -2; done
LIVO2_FENCE_GRAN=recommended LIVO2_REDZONE=2 timeout 120 $S oob > $O/selftest_rec_oob.txt 2>&1; echo "oob rc=$?"
cd /tmp && /opt/rocm/bin/hipcc --offload-arch=gfx950:xnack+ -fsanitize=address -shared-libsan -g $GRAFT_REPO_ROOT/tools/asan_probe.hip -o asan_probe > $GRAFT_REPO_ROOT/$O/asan.txt 2>&1
LD_LIBRARY_PATH=/opt/rocm/lib/llvm/lib/clang/22/lib/linux:/opt/rocm/lib HSA_XNACK=1 timeout 60 ./asan_probe >> $GRAFT_REPO_ROOT/$O/asan.txt 2>&1; echo "asan rc=$?"; tail -12 $GRAFT_REPO_ROOT/$O/asan.txt | cut -c1-200
