#!/usr/bin/env python
"""Block-size experiment for the single-scan LiDAR residual kernel on a GPU box: the C2 scan (or, with `c4`, the bench's C4 frame) and a 17k-point scan with the
block size forced through the environment (LIVO2_LIDAR_BLOCK=128|256; unset = the library's own choice by scan size) or through variant builds
(fast-livo2_amd/lib/liblivo2_hip_<suffix>.so): per-iteration wall time, HIP-event kernel time of residual and solve, full-update latency.
Usage: [LIVO2_LIDAR_BLOCK=128] python tools/block_probe.py [c4] [lib suffixes]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402
import importlib as _il  # noqa: E402
H = _il.import_module("fast-livo2_amd.configs")  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
abi = livo2.abi
base = abi.LIB_PATH
args = [a for a in sys.argv[1:] if a != "c4"]
if "c4" in sys.argv[1:]:                      # the bench's C4 frame (200 000 post-filter points)
    import bench
    sc = bench.c4_frame(4, 200000, 4000)[0]
else:
    sc = synth.lidar_scenario(seed=2, n_points=100000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=12, downsample=synth.AVIA["filter_size_surf"])
small = synth.lidar_scenario(seed=1, n_points=24000, downsample=0.1)
ref_state = None
for suffix in (args or [""]):
    path = base if not suffix else base.replace(".so", "_" + suffix + ".so")
    if not os.path.exists(path):
        print(suffix, "missing", path); continue
    abi._lib, abi.LIB_PATH = None, path
    ctx = livo2.Context(0)
    row = [(suffix or "lib") + " block=" + os.environ.get("LIVO2_LIDAR_BLOCK", "auto")]
    for s in (sc, small):
        cfg = H.lidar_cfg(s)
        ctx.upload_map(s.fmap); ctx.set_scan(s.xyz, cfg)
        cur, prop = H.prior_states(s)
        ctx.lidar_iterations_async(cur, prop, cfg, 20); ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); ctx.lidar_iterations_async(cur, prop, cfg, 200); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
        ctx.kernel_timing(True); ctx.kernel_timing_read(0); ctx.kernel_timing_read(2)
        ctx.lidar_iterations_async(cur, prop, cfg, 200)
        ms_res, n_res = ctx.kernel_timing_read(0); ms_sol, n_sol = ctx.kernel_timing_read(2)
        ctx.kernel_timing(False)
        tf = 1e9
        for _ in range(20):
            t0 = time.perf_counter(); res, _ = ctx.lidar_update(cur, prop, cfg); tf = min(tf, time.perf_counter() - t0)
        row += [len(s.xyz), f"iter_us={best * 1e6:.2f}", f"res_us={1e3 * ms_res / n_res:.2f}", f"sol_us={1e3 * ms_sol / n_sol:.2f}", f"full_ms={tf * 1e3:.4f}", f"iters={res.n_iters}"]
        if s is sc:
            v = np.frombuffer(bytes(res.state), np.float64)[:12]
            ref_state = v if ref_state is None else ref_state
            row.append(f"dstate={np.abs(v - ref_state).max():.1e}")
    print(*row, flush=True)
    ctx.close()
