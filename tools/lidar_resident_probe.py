#!/usr/bin/env python
"""k_lidar_residual with a resident grid (LIVO2_LIDAR_RESIDENT blocks looping over the scan's chunks) against one block per chunk: event time per launch at C4 / C2.
Usage (GPU box): LIVO2_LIDAR_RESIDENT=<0|512|...> python tools/lidar_resident_probe.py"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")


def main():
    sc, vs = bench.c4_frame(4, 200000, 4000)
    ctx = livo2.Context(0)
    cfg = H.lidar_cfg(sc)
    ctx.upload_map(sc.fmap)
    for n in (200000, 100000, 17000):
        ctx.set_scan(sc.xyz[:n], cfg)
        cur, prop = H.prior_states(sc)
        res, _ = ctx.lidar_update(cur, prop, cfg)
        ctx.lidar_iterations_async(cur, prop, cfg, 20); ctx.synchronize()
        ctx.kernel_timing(True)
        for b in range(4):
            ctx.kernel_timing_read(b)
        ctx.lidar_iterations_async(cur, prop, cfg, 200); ctx.synchronize()
        r, s = ctx.kernel_timing_read(0), ctx.kernel_timing_read(2)
        ctx.kernel_timing(False)
        print(f"resident={os.environ.get('LIVO2_LIDAR_RESIDENT', 'default')} n={n} iters={res.n_iters} residual {1e3 * r[0] / r[1]:.2f} us  solve {1e3 * s[0] / s[1]:.2f} us  state t={list(res.state.pos)}", flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
