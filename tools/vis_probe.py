#!/usr/bin/env python
"""Visual residual / solve kernel probe on a GPU box: the bench's C4 visual frame (4 000 patches, level 0) through liblivo2_hip.so and variant builds
(hipcc ... -DVIS_EXP_... -o fast-livo2_amd/lib/liblivo2_hip_<suffix>.so): wall time per (residual + solve) step and HIP-event kernel times.
Usage: python tools/vis_probe.py [lib suffixes, default '']"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402
import importlib as _il  # noqa: E402
H = _il.import_module("fast-livo2_amd.configs")  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
abi = livo2.abi
base = abi.LIB_PATH
vs = synth.visual_scenario(seed=5, n_patches=4000)
for suffix in (sys.argv[1:] or [""]):
    path = base if not suffix else base.replace(".so", "_" + suffix + ".so")
    if not os.path.exists(path):
        print(suffix, "missing", path); continue
    abi._lib, abi.LIB_PATH = None, path
    ctx = livo2.Context(0)
    cfg = H.visual_cfg(vs, mp_proc_num=4)
    cur, prop = H.prior_states(vs)
    ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    row = [suffix or "base"]
    for level in (0, 3):
        ctx.visual_iterations_async(level, cur, prop, cfg, 20); ctx.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); ctx.visual_iterations_async(level, cur, prop, cfg, 200); ctx.synchronize(); best = min(best, (time.perf_counter() - t0) / 200)
        ctx.kernel_timing(True); ctx.kernel_timing_read(1); ctx.kernel_timing_read(3)
        ctx.visual_iterations_async(level, cur, prop, cfg, 200)
        ms_res, n_res = ctx.kernel_timing_read(1); ms_sol, n_sol = ctx.kernel_timing_read(3)
        ctx.kernel_timing(False)
        row += [f"L{level}: step_us={best * 1e6:.2f} res_us={1e3 * ms_res / n_res:.2f} sol_us={1e3 * ms_sol / max(n_sol, 1):.2f}"]
    tf = 1e9
    for _ in range(10):
        t0 = time.perf_counter(); res, _ = ctx.visual_update(cur, prop, cfg); tf = min(tf, time.perf_counter() - t0)
    row.append(f"full_update_ms={tf * 1e3:.4f} steps={res.n_steps}")
    print(*row, flush=True)
    ctx.close()
