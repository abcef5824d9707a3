"""extra.visual_inverse of the bench alone (torch-free): python tools/inverse_probe.py [patches=2000]"""
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
out = bench_legs._visual_inverse_leg(ctx, livo2, synth, H, int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
for k, v in out.items():
    print(k, json.dumps(v) if isinstance(v, dict) else v)
