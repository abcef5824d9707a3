#!/usr/bin/env python
"""Per-phase latency of k_visual_residual from the profiling build (make -C fast-livo2_amd/csrc prof): C4 visual frame (4 000 patches), level 0.
--persistent: the residual phase of k_visual_update_persistent instead (stamps of the last step of a whole update; slots 0 -> 6)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

livo2 = importlib.import_module("fast-livo2_amd")
livo2.abi.LIB_PATH = os.path.join(ROOT, "fast-livo2_amd", "lib", "liblivo2_hip_prof.so")
from scenarios import synth  # noqa: E402
import importlib as _il  # noqa: E402
H = _il.import_module("fast-livo2_amd.configs")  # noqa: E402

PERSISTENT = "--persistent" in sys.argv
if PERSISTENT:
    sys.argv.remove("--persistent")
vs = synth.visual_scenario(seed=5, n_patches=int(sys.argv[1]) if len(sys.argv) > 1 else 4000)
ctx = livo2.Context(0)
cfg = H.visual_cfg(vs, mp_proc_num=4)
cur, prop = H.prior_states(vs)
ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
if PERSISTENT:
    for _ in range(5):
        ctx.visual_update(cur, prop, cfg)
else:
    ctx.visual_iterations_async(0, cur, prop, cfg, 20); ctx.synchronize()
    ctx.visual_iterations_async(0, cur, prop, cfg, 5); ctx.synchronize()
fn = ctx.lib.livo2_debug_vis_prof
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
W = (len(vs.pos) + 3) // 4
W = (W + 7) // 8 * 8
full = np.zeros((1 << 14, 8), np.uint64)
assert fn(ctx.h, full.ctypes.data_as(C.POINTER(C.c_uint64)), 1 << 14) == 0
buf = full[:W]
st = buf.astype(np.int64)
st = st[st[:, 6] > 0]
t0 = st[:, 0].min()
start, end = (st[:, 0] - t0) * 0.01, (st[:, 7] - t0) * 0.01
print(f"{len(st)} waves; chip clock (us): wave start p5 {np.percentile(start, 5):.2f} p50 {np.percentile(start, 50):.2f} p95 {np.percentile(start, 95):.2f} max {start.max():.2f}; "
      f"wave end p5 {np.percentile(end, 5):.2f} p50 {np.percentile(end, 50):.2f} p95 {np.percentile(end, 95):.2f} max {end.max():.2f}; life p50 {np.percentile(end - start, 50):.2f} max {(end - start).max():.2f}")
if PERSISTENT:
    names = ["0->1 pos / search_level / inv_expo / reference pixels from global memory", "1->2 Rcw, pf, projection, window loads issued+drained", "2->3 weights, Jpi, M, Wf, B grid", "3->4 pixel loop",
             "4->5 tile sums + float chain", "5->6 expansion + error store"]
    cyc = np.median((st[:, 6] - st[:, 1])[st[:, 6] > st[:, 1]])
    for k, n in enumerate(names):
        d = (st[:, k + 1] - st[:, k]) / 2400.0 if k else np.zeros(len(st))
        print(f"{n:85s} mean {d.mean():6.2f} p50 {np.percentile(d, 50):6.2f} p95 {np.percentile(d, 95):6.2f} us (cycles/2400)")
    d = (st[:, 6] - st[:, 1]) / 2400.0
    print(f"{'1->6':85s} mean {d.mean():6.2f} p50 {np.percentile(d, 50):6.2f} p95 {np.percentile(d, 95):6.2f} us; raw counter units p50 {cyc:.0f}")
    ctx.close()
    sys.exit(0)
names = ["1->2 state arrives, Rcw, pf, projection, window loads issued+drained", "2->3 weights, Jpi, M, Wf, B grid", "3->4 pixel loop", "4->5 tile sums + float chain", "5->6 expansion"]
for k, n in enumerate(names):
    d = (st[:, k + 2] - st[:, k + 1]) / 2100.0
    print(f"{n:75s} mean {d.mean():6.2f} p50 {np.percentile(d, 50):6.2f} p95 {np.percentile(d, 95):6.2f} us (cycles/2100)")
d = (st[:, 6] - st[:, 1]) / 2100.0
print(f"{'1->6 whole body':75s} mean {d.mean():6.2f} p50 {np.percentile(d, 50):6.2f} p95 {np.percentile(d, 95):6.2f} us")
sv = full.astype(np.int64)
for w, name in ((0, 'wave 0 (algebra)'), (2, 'wave 2 (error chain)'), (1, 'wave 1 (Log)')):
    r = sv[(1 << 14) - 1 - w]
    ph = [(r[k + 1] - r[k]) / 2100.0 for k in range(1, 5)]
    print(f"k_visual_solve {name}: starts {(r[0] - t0) * 0.01:.2f} us after the residual kernel's first wave, runs {(r[7] - r[0]) * 0.01 if r[7] else float('nan'):.2f} us; "
          f"loads issued+landed {ph[0]:.2f}, barrier {ph[1]:.2f}, sums {ph[2]:.2f}, algebra-or-chain {ph[3]:.2f}" + (f", barrier {(r[5] - r[4]) / 2100.0:.2f}, decide+commit {(r[6] - r[5]) / 2100.0:.2f}" if w == 0 else ''))
ctx.close()
