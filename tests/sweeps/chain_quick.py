#!/usr/bin/env python
"""Quick look at the chained retrieval on a GPU box: kernel time of livo2_visual_retrieve_from_map (avia grid, 30k visual points) next to the
single-thread oracle, for both reference-patch choice modes."""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402  (checker / CPU baseline only)
from scenarios import synth  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
ctx = livo2.Context(0)
for normal_en in (True, False):
    cs = synth.retrieve_chain_scenario(seed=81, n_pg=10000, n_vis=30000, grid_n_height=102, normal_en=normal_en)
    t0 = time.perf_counter(); ref = orc.visual_retrieve(cs); cpu = time.perf_counter() - t0
    ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
    us = []
    for _ in range(6):
        ctx.visual_obs_upload(cs)
        out = ctx.visual_retrieve_from_map(cs, want_patches=False)
        us.append(ctx.retrieve_from_map_last_kernel_us())
    same = np.array_equal(out["cell_obs"], ref["cell_obs"]) and np.array_equal(out["sub_point"], ref["sub_point"]) and np.array_equal(out["tail"]["error"], ref["tail"]["error"])
    print(f"normal_en={int(normal_en)} cells={len(out['cell_point'])} candidates={out['n_candidates']} accepted={out['n_accepted']} kernel_us={np.median(us[1:]):.1f} "
          f"(first {us[0]:.1f}) oracle_1thread_ms={cpu * 1e3:.2f} identical={same}", flush=True)
ctx.close()
