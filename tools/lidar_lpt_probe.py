#!/usr/bin/env python
"""Longest-lifetime-first block order of k_lidar_residual (option "lidar_block_order": every launch records per-chunk block lifetimes, an idle wave of the solve sorts
them into the launch order of the next iteration) against the identity order.  Event time per launch, C4 scan cut to three sizes.
Usage (GPU box): python tools/lidar_lpt_probe.py > gpurun_out/lidar_lpt_probe.txt"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")


def timed(ctx, cur, prop, cfg, reps=200):
    ctx.lidar_iterations_async(cur, prop, cfg, 20); ctx.synchronize()
    ctx.kernel_timing(True)
    for b in range(4):
        ctx.kernel_timing_read(b)
    ctx.lidar_iterations_async(cur, prop, cfg, reps); ctx.synchronize()
    r, s = ctx.kernel_timing_read(0), ctx.kernel_timing_read(2)
    ctx.kernel_timing(False)
    return 1e3 * r[0] / r[1], 1e3 * s[0] / s[1]


def main():
    sc, vs = bench.c4_frame(4, 200000, 4000)
    ctx = livo2.Context(0)
    cfg = H.lidar_cfg(sc)
    ctx.upload_map(sc.fmap)
    for n in (200000, 100000, 17000):
        out = []
        for on in (0, 1, 0, 1):
            ctx.set_option("lidar_block_order", on)
            ctx.set_scan(sc.xyz[:n], cfg)
            cur, prop = H.prior_states(sc)
            res, _ = ctx.lidar_update(cur, prop, cfg)
            out.append((on, timed(ctx, cur, prop, cfg), bytes(res.state)))
        assert all(o[2] == out[0][2] for o in out)
        print(f"n={n}: " + " | ".join(f"order {'LPT' if o[0] else 'identity'}: residual {o[1][0]:.2f} us solve {o[1][1]:.2f} us" for o in out) + " | identical result bits", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
