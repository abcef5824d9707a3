#!/usr/bin/env python
"""Where a C1-shaped C5 frame spends its 0.6 ms (host-synchronous calls through the Python wrappers): per-call wall times of frames.run_frame's four calls, and the
frame rate with 1 / 2 / 3 contexts.  Usage (GPU box): python tools/c5_stage_probe.py > gpurun_out/c5_stage_probe.txt"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
frames_mod = importlib.import_module("fast-livo2_amd.frames")
cfgs = importlib.import_module("fast-livo2_amd.configs")


def main():
    fmap, lio_cfg, extR, extT, frames = bench.c5_frames(64, "c1")
    cfg = cfgs.lidar_cfg(bench._Sc(lio_cfg, extR, extT)); vcfg = cfgs.visual_cfg(frames[0]["vs"], mp_proc_num=4)
    ctxs = [livo2.Context(0) for _ in range(3)]
    for c in ctxs:
        c.upload_map(fmap)
        frames_mod.run_frame(c, livo2.State, frames[0], cfg, vcfg)
    c = ctxs[0]
    t = np.zeros(5)
    for f in frames:
        prior = livo2.State.from_pose(f["R_prior"], f["t_prior"], f["P"])
        vs = f["vs"]
        vprior = livo2.State.from_pose(vs.R_prior, vs.t_prior, vs.P, inv_expo=getattr(vs, "tau_prior", 1.0))
        t0 = time.perf_counter(); c.set_scan(f["xyz"], cfg)
        t1 = time.perf_counter(); lres, _ = c.lidar_update(prior, prior, cfg)
        t2 = time.perf_counter(); c.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        t3 = time.perf_counter(); vres, _ = c.visual_update(vprior, vprior, vcfg)
        t4 = time.perf_counter(); frames_mod.pack_result(lres, vres)
        t5 = time.perf_counter()
        t += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4]
    t *= 1e6 / len(frames)
    print("us per frame: set_scan %.0f  lidar_update %.0f  set_frame %.0f  visual_update %.0f  pack_result %.0f  sum %.0f" % (*t, t.sum()))
    for K in (1, 2, 3):
        t0 = time.perf_counter()
        frames_mod.run_frames_sharded(ctxs[:K] if K > 1 else ctxs[0], livo2.State, frames, cfg, vcfg, 0, 1)
        for x in ctxs[:K]:
            x.synchronize()
        dt = time.perf_counter() - t0
        print("contexts %d: %.0f frames/s" % (K, len(frames) / dt))


if __name__ == "__main__":
    main()
