"""livo2_lidar_preprocess_scan alone (torch-free): 240 000 raw points as in extra.preprocess_scan, and an avia-sized scan (24 000).  python tools/prestage_probe.py"""
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")
ctx = livo2.Context(0)
sc = synth.lidar_scenario(seed=1, n_points=2000, downsample=0.1)
cfg = H.lidar_cfg(sc)
for n_raw in (24000, 240000):
    raw = synth.raw_scan_scenario(seed=1, n_raw=n_raw)
    for _ in range(3):
        nd = ctx.preprocess_scan(raw.xyz, raw.curvature, raw.poses, raw.rot_end, raw.pos_end, raw.leaf, cfg, want=False)[0]
    ts, ks = [], []
    for _ in range(10):
        t0 = time.perf_counter(); ctx.preprocess_scan(raw.xyz, raw.curvature, raw.poses, raw.rot_end, raw.pos_end, raw.leaf, cfg, want=False); ts.append(time.perf_counter() - t0)
        ks.append(ctx.lib.livo2_lidar_preprocess_last_kernel_us(ctx.h))
    print("raw %d -> %d leaves: call %.1f us (min), kernel span by events %.1f us (min)" % (n_raw, nd, 1e6 * min(ts), min(ks)))
