#!/usr/bin/env python
"""Persistent visual update (k_visual_update_persistent) against the launch-per-step sequence: wall time per computeJacobianAndUpdateEKF, one update in flight.
Usage (GPU box): python tools/vis_persist_probe.py [M] [reps] > gpurun_out/vis_persist_probe.txt"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")


def cfg_of(vs, mp=4):
    c = livo2.VisualCfg()
    c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"]
    c.cam.distortion, c.cam.width, c.cam.height = 0, vs.cam["width"], vs.cam["height"]
    c.Rcl[:] = vs.Rcl.ravel().tolist(); c.Pcl[:] = vs.Pcl.tolist(); c.extR[:] = vs.extR.ravel().tolist(); c.extT[:] = vs.extT.tolist()
    c.img_point_cov = float(vs.cfg["img_point_cov"]); c.patch_pyrimid_level = int(vs.cfg["patch_pyrimid_level"]); c.max_iterations = int(vs.cfg["max_iterations"])
    c.exposure_estimate_en, c.inverse_composition_en, c.mp_proc_num = 1, 0, mp
    return c


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    ctx = livo2.Context(0)
    ew = os.environ.get("LIVO2_PROBE_ERROR_WAVES")                 # "0": the frame error on one lane per thread (option "visual_error_waves")
    if ew is not None:
        ctx.set_option("visual_error_waves", int(ew))
    for seed in (4, 5):
        vs = synth.visual_scenario(seed=seed, n_patches=M)
        cfg = cfg_of(vs)
        prior = livo2.State.from_pose(vs.R_prior, vs.t_prior, vs.P, inv_expo=getattr(vs, "tau_prior", 1.0))
        ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        only = os.environ.get("LIVO2_PROBE_ONLY", "")            # "per-step": experiments with builds whose persistent kernel is not valid
        for name, opt in (("per-step", 0), ("persistent", 1), ("per-step", 0), ("persistent", 1)):
            if only and name != only:
                continue
            ctx.set_option("visual_persistent", opt)
            res, _ = ctx.visual_update(prior, prior, cfg)
            for _ in range(20):
                ctx.visual_update_async(prior, prior, cfg)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.visual_update_async(prior, prior, cfg)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(f"seed {seed} M {M} {name:10s} steps {res.n_steps:2d}  {dt * 1e6:8.1f} us per update  ({dt * 1e6 / max(res.n_steps, 1):6.2f} us per executed step)", flush=True)
    print("persistent launches", ctx.counter("visual_persistent_launches"), "fallbacks", ctx.counter("visual_persistent_fallbacks"))
    ctx.close()
    if only:
        return
    # phase stamps of one persistent update (100 MHz clock): 0 step start, 1 residual done, 2 row stored, 3 barrier passed, 4 rows + errors in LDS, 5 solve + chain done, 6 decision
    import ctypes as C
    import numpy as np
    os.environ["LIVO2_VP_PROF"] = "1"
    c2 = livo2.Context(0)
    del os.environ["LIVO2_VP_PROF"]
    if ew is not None:
        c2.set_option("visual_error_waves", int(ew))
    c2.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
    for _ in range(3):
        res, _ = c2.visual_update(prior, prior, cfg)
    buf = np.zeros((256, 32, 16), np.uint64)
    c2.lib.livo2_debug_vp_prof.argtypes = [C.c_void_p, C.c_void_p]
    assert c2.lib.livo2_debug_vp_prof(c2.h, buf.ctypes.data_as(C.c_void_p)) == 0
    t00 = int(buf[0, 0, 0])
    print("# block step : start  residual  store  barrier  loads  solve  decide   (us, phase durations; start relative to block 0 step 0)")
    for b in (0, 3, 7):
        for st in range(res.n_steps):
            t = buf[b, st].astype(np.int64)
            d = [(int(t[k + 1]) - int(t[k])) / 100.0 for k in range(6)]
            print(f"  {b} {st:2d} : {(int(t[0]) - t00) / 100.0:7.2f}  " + "  ".join(f"{x:6.2f}" for x in d) + f"   from sums: chain_end {(int(t[7]) - int(t[4])) / 100.0:5.2f} hth {(int(t[8]) - int(t[4])) / 100.0:5.2f} solve_end {(int(t[9]) - int(t[4])) / 100.0:5.2f}")
    # all blocks: when does a block finish its residual / see every word, relative to block 0's step start (which blocks make the others wait?)
    G = int((buf[:, 0, 0] != 0).sum())
    print(f"# {G} blocks; per step: residual-end offsets (us, vs block 0's step start) min / median / p90 / max ; all-words-arrived min / max ; slowest 6 blocks (by residual end)")
    for st in range(min(res.n_steps, 6)):
        s0 = int(buf[0, st, 0])
        r_end = (buf[:G, st, 1].astype(np.int64) - s0) / 100.0
        start = (buf[:G, st, 0].astype(np.int64) - s0) / 100.0
        arr = (buf[:G, st, 3].astype(np.int64) - s0) / 100.0
        slow = np.argsort(r_end)[-6:][::-1]
        print(f"  step {st}: start {start.min():6.2f}..{start.max():6.2f}  residual end {r_end.min():6.2f} / {np.median(r_end):6.2f} / {np.percentile(r_end, 90):6.2f} / {r_end.max():6.2f} ; arrived {arr.min():6.2f} .. {arr.max():6.2f} ; slowest " + " ".join(f"{b}:{r_end[b]:.2f}" for b in slow))
    c2.close()


if __name__ == "__main__":
    main()
