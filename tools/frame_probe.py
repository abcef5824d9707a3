"""Where a C1-shaped frame's time goes through livo2_frame_update (torch-free): host time of the _async call alone, frames/s with two frames in flight on one
context over PREPARED livo2_frame_in structs, and with K contexts on K host threads.   python tools/frame_probe.py [c1|c4] [frames] [contexts]"""
import ctypes as C
import importlib
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    shape = sys.argv[1] if len(sys.argv) > 1 else "c1"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    livo2 = importlib.import_module("fast-livo2_amd")
    H = importlib.import_module("fast-livo2_amd.configs")
    F = importlib.import_module("fast-livo2_amd.frames")
    fmap, lio_cfg, extR, extT, distinct = bench.c5_frames(min(n, 64 if shape == "c1" else 12), shape)
    frames = [distinct[f % len(distinct)] for f in range(n)]
    cfg = H.lidar_cfg(bench._Sc(lio_cfg, extR, extT)); vcfg = H.visual_cfg(frames[0]["vs"], mp_proc_num=4)
    ctxs = [livo2.Context(0) for _ in range(K)]
    for c in ctxs:
        c.upload_map(fmap)
        F.run_frame(c, livo2.State, frames[0], cfg, vcfg)
    c0 = ctxs[0]
    prep = F.prepare_frames(c0, livo2.State, frames, cfg, vcfg)
    F.run_prepared(c0, frames, prep[:8]); c0.synchronize()
    # host time of the enqueue alone (the stream drains in between)
    ts = []
    for k in range(16):
        t0 = time.perf_counter(); c0._chk(c0.lib.livo2_frame_update_async(c0.h, C.byref(prep[k][1]))); ts.append(time.perf_counter() - t0)
        c0._chk(c0.lib.livo2_frame_update_fetch(c0.h, C.byref(prep[k][3]), C.byref(prep[k][4])))
    print("host time of livo2_frame_update_async: median %.1f us (min %.1f)" % (1e6 * np.median(ts), 1e6 * min(ts)))
    ts = []
    for k in range(16):
        t0 = time.perf_counter(); c0._chk(c0.lib.livo2_frame_update(c0.h, C.byref(prep[k][1]), C.byref(prep[k][3]), C.byref(prep[k][4]))); ts.append(time.perf_counter() - t0)
    print("latency of one synchronous livo2_frame_update: median %.1f us" % (1e6 * np.median(ts)))
    for rep in range(2):
        c0.synchronize(); t0 = time.perf_counter(); recs, _ = F.run_prepared(c0, frames, prep); dt = time.perf_counter() - t0
        print("one context, two frames in flight: %.0f frames/s (%.3f ms per frame)" % (n / dt, 1e3 * dt / n))
    preps = [F.prepare_frames(ctxs[j], livo2.State, frames, cfg, vcfg, list(range(j, n, K))) for j in range(K)]
    for rep in range(5):
        out = [None] * K
        if rep == 3:                                   # another order of the same frames: is the first pass slow because of the ORDER of sizes, or once only?
            preps = [list(reversed(p)) for p in preps]

        def work(j):
            out[j] = F.run_prepared(ctxs[j], frames, preps[j])
        th = [threading.Thread(target=work, args=(j,)) for j in range(K)]
        t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; dt = time.perf_counter() - t0
        print("%d contexts / host threads, two frames in flight each: %.0f frames/s" % (K, n / dt))
    allrec = np.zeros_like(recs)
    for j in range(K):
        allrec[j::K] = out[j][0][::-1]
    print("records of the two runs equal:", bool(np.array_equal(allrec, recs)))
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
