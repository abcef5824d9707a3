"""extra.map_update of the bench alone (torch-free): kernel and call time of livo2_map_tree_update_from_scan per frame at avia size and at 96 k points.  python tools/map_probe.py"""
import importlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenarios import synth  # noqa: E402
from tools import bench_legs  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
H = importlib.import_module("fast-livo2_amd.configs")
ctx = livo2.Context(0)
extra = {}
bench_legs._map_update_leg(ctx, livo2, synth, H, extra)
for k, v in extra["map_update"].items():
    if isinstance(v, dict):
        print(k, "call ms median %.3f  kernel us median %.1f  per frame kernel us %s" % (v["map_update_ms_median"], v["map_update_kernel_us_median"], [round(f["map_update_kernel_us"], 1) for f in v["frames"]]))
