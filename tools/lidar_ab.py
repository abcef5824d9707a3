"""A/B of LiDAR launch options on the bench's own C4 workload, in ONE process on one box and without torch (a fresh box pays 1-2 min for its first
`import torch`; this tool is ctypes + numpy only).  Prints per variant: k_lidar_residual / k_lidar_solve by HIP events (the library's own event pass,
livo2_ctx_kernel_timing) and the wall time of 8 frame updates without events.  Variants are visited round-robin `--rounds` times so that clock drift of the box
shows up as spread inside a variant, not as a difference between variants.

python tools/lidar_ab.py [--rounds 3] [--variants order=1 order=0]      (a variant is a comma list of <livo2_ctx_set_option name>=<value>; order = lidar_block_order;
name EVERY option in every variant: a value set by one variant stays until another one changes it)
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (torch is imported inside bench.main only)


def device_clock(ctx):
    try:
        return ctx.counter("lidar_residual_device_ticks"), ctx.counter("lidar_residual_device_launches")
    except Exception:
        return 0, 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--points", type=int, default=200000)
    ap.add_argument("--variants", nargs="*", default=["order=1", "order=0"])
    args = ap.parse_args()
    from scenarios import synth
    livo2 = importlib.import_module("fast-livo2_amd")
    H = importlib.import_module("fast-livo2_amd.configs")
    sc, vs = bench.c4_frame(4, args.points, 4000)
    ctx = livo2.Context(0)
    w = bench.C4(ctx, livo2, synth, H, sc, vs, 8, seed=100)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        w.run(4); ctx.synchronize()
    n_lid = sum(w.iters)
    rows = {v: [] for v in args.variants}
    for rnd in range(args.rounds):
        for v in args.variants:
            for kv in v.split(","):
                k, val = kv.split("=")
                ctx.set_option({"order": "lidar_block_order"}.get(k, k), int(val))
            w.run(3); ctx.synchronize()
            t1 = time.perf_counter(); w.run(args.steps); ctx.synchronize(); wall = (time.perf_counter() - t1) / args.steps
            ctx.kernel_timing(True)
            for b in range(4):
                ctx.kernel_timing_read(b)
            dev0 = device_clock(ctx)
            w.run(args.steps); ctx.synchronize()
            bins = [ctx.kernel_timing_read(b) for b in range(4)]
            dev1 = device_clock(ctx)
            ctx.kernel_timing(False)
            dev_us = 0.01 * (dev1[0] - dev0[0]) / max(dev1[1] - dev0[1], 1)        # the kernel's own span on the 100-MHz device clock (round 6; 0 with a library that has no stamps)
            res_us, sol_us = 1e3 * bins[0][0] / (n_lid * args.steps), 1e3 * bins[2][0] / (n_lid * args.steps)
            vis_us = 1e3 * bins[1][0] / (len(w.vsteps) * args.steps)
            # LiDAR updates only, no events, no profiler: wall time per executed iteration (residual + solve + launch gaps) — an event record or a profiler's
            # counter packets between two launches may carry fences that an uninstrumented stream does not
            for f in range(w.F):
                ctx.lidar_update_async(w.lid[f], w.lid[f], w.cfg)
            ctx.synchronize()
            t2 = time.perf_counter()
            for _ in range(args.steps * 4):
                for f in range(w.F):
                    ctx.lidar_update_async(w.lid[f], w.lid[f], w.cfg)
            ctx.synchronize()
            it_us = 1e6 * (time.perf_counter() - t2) / (n_lid * args.steps * 4)
            rows[v].append((res_us, sol_us, vis_us, 1e3 * wall, it_us, dev_us))
            print(f"round {rnd} {v:16s} k_lidar_residual {res_us:6.2f} us (device clock {dev_us:6.2f})  k_lidar_solve {sol_us:6.2f} us  visual update {vis_us:7.1f} us  8 frames {1e3 * wall:6.3f} ms  "
                  f"lidar-only wall per iteration {it_us:6.2f} us", flush=True)
    print("# medians")
    for v in args.variants:
        m = np.median(np.array(rows[v]), axis=0)
        print(f"{v:16s} k_lidar_residual {m[0]:6.2f} us (device clock {m[5]:6.2f})  k_lidar_solve {m[1]:6.2f} us  visual update {m[2]:7.1f} us  8 frames {m[3]:6.3f} ms  lidar-only wall per iteration {m[4]:6.2f} us")


if __name__ == "__main__":
    main()
