#!/bin/bash
# copies one tools/capture_profiles.sh evidence set from gpurun_out/prof_<tag>/ to the names profiles/ keeps: tools/adopt_capture.sh r06
T=${1:-r06}; P=gpurun_out/prof_$T
cp $P/bench.json profiles/${T}_bench_line.json; cp $P/bench_full.json profiles/${T}_bench_full.json; cp $P/build_sha.txt profiles/${T}_build_sha.txt
cp $P/frame_api_probe.txt profiles/${T}_frame_api_probe.txt
cp $P/kernel_trace_stats_c1_frame.txt profiles/${T}_rocprofv3_kernel_trace_stats_c1_frame.txt; cp $P/kernel_trace_stats_c4.txt profiles/${T}_rocprofv3_kernel_trace_stats_c4.txt
cat $P/pmc_c4_lockstep.txt $P/pmc_batched.txt $P/pmc_out_of_cache.txt > profiles/${T}_rocprofv3_pmc_batched_legs.txt
cp $P/pmc_c4.txt profiles/${T}_rocprofv3_pmc_c4.txt; cp $P/pmc_c4_waves.txt profiles/${T}_rocprofv3_pmc_c4_waves.txt
cp $P/${T}_traffic_*.json profiles/; cp $P/parity_sweep.txt profiles/${T}_parity_sweep.txt; cp $P/pytest_gpu.txt profiles/${T}_pytest_gpu.txt
cp $P/vis_persist_probe.txt profiles/${T}_visual_persistent_probe.txt
