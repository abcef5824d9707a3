#!/usr/bin/env python
"""Per-phase latency breakdown of k_lidar_residual from the profiling build (make -C fast-livo2_amd/csrc prof).
Loads liblivo2_hip_prof.so in place of the product library, runs a few ESIKF iterations on the C2 scenario and prints, per phase,
the mean / p50 / p95 / max wave time in microseconds (s_memtime ticks at 100 MHz on gfx950 -> 10 ns per tick)."""
import ctypes as C
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    order = sys.argv[1] if len(sys.argv) > 1 else "voxelgrid"
    livo2 = importlib.import_module("fast-livo2_amd")
    livo2.abi.LIB_PATH = os.path.join(ROOT, "fast-livo2_amd", "lib", "liblivo2_hip_prof.so")
    from scenarios import synth
    import importlib as _il
    H = _il.import_module("fast-livo2_amd.configs")
    import bench
    from tools import bench_legs
    if order == "c4":
        sc = bench.c4_frame(4, 200000, 4000)[0]
    else:
        sc = synth.lidar_scenario(seed=2, n_points=100000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=12,
                                  downsample=(0.1 if order == "voxelgrid" else None))
    ctx = livo2.Context(0)
    cfg = H.lidar_cfg(sc)
    ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
    cur, prop = bench_legs.make_states(livo2, sc)
    ctx.lidar_iterations_async(cur, prop, cfg, 10); ctx.synchronize()
    ctx.lidar_iterations_async(cur, prop, cfg, 5); ctx.synchronize()
    fn = ctx.lib.livo2_debug_phase_prof
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
    W = 1 << 17
    buf = np.zeros((W, 8), np.uint64)
    nw = C.c_size_t()
    assert fn(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), W, C.byref(nw)) == 0
    waves = ((len(sc.xyz) + 255) // 256 + 7) // 8 * 8 * 4        # the launch's own grid (256-point blocks, a multiple of 8): the stamp regions are laid out by gridDim
    cst = buf[waves + 2 + waves // 4:waves + 2 + waves // 4 + 2 * waves].reshape(waves, 16).astype(np.int64)   # stamps inside the first cooperative round
    rt = buf[waves + 2:waves + 2 + waves // 4].reshape(-1)[:2 * waves].reshape(waves, 2).astype(np.int64)   # chip-wide 100 MHz stamps
    sol = buf[waves].astype(np.int64)
    sol_rt = buf[waves + 1, :2].astype(np.int64)
    nw.value = waves
    print('solve kernel phases (us @2.1GHz): ' + ', '.join(f'{n} {(sol[k + 1] - sol[k]) / 2100.0:.2f}' for k, n in enumerate(['prefetch-issue', 'partial reduce', 'sums out', 'vec+GJ+gain+state', 'conv/cov/ctl'])))
    st = buf[:nw.value].astype(np.int64)
    st = st[st[:, 0] > 0]
    # stamps are per-XCD counters: normalise each XCD (block b runs on XCD b % 8) to its own first wave
    xcd = (np.arange(len(buf[:nw.value])) // 4 % 8)[buf[:nw.value, 0] > 0]
    for x in range(8):
        m = xcd == x
        if m.any():
            st[m, :7] -= st[m, 0].min()
    names = ["launch skew (first wave start -> this wave start)", "T1 xyz+cov+state, transform, key", "T2 cuckoo slots (+ sym P)", "T3 first visit (plane / candidates)",
             "T4 neighbour visit", "accumulate + outputs", "LDS reduction + partial store"]
    tick_us = 0.01
    t0 = st[:, 0].min()
    rows = [("0 " + names[0], (st[:, 0] - t0) * tick_us)]
    for k in range(1, 7):
        ok = st[:, k] > 0
        d = np.where(ok, st[:, k] - st[:, k - 1], 0)
        # phases 3/4 are skipped by waves whose lanes never reach them: carry the previous stamp forward
        if not ok.all():
            st[:, k] = np.where(ok, st[:, k], st[:, k - 1])
        rows.append((f"{k} " + names[k], d * tick_us))
    print(f"scan order {order}: {len(st)} waves, kernel span {(st[:, 6].max() - t0) * tick_us:.2f} us")
    for n, d in rows:
        print(f"{n:55s} mean {d.mean():6.2f}  p50 {np.percentile(d, 50):6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f} us")
    W = st[:, 7]
    d3 = (st[:, 3] - st[:, 2]) * tick_us
    for lo, hi in ((0, 0), (1, 8), (9, 32), (33, 64), (65, 10**9)):
        m = (W >= lo) & (W <= hi)
        if m.any():
            print(f"  waves with {lo}..{hi} candidate pairs: {m.sum():5d}  T3 mean {d3[m].mean():7.2f} p95 {np.percentile(d3[m], 95):7.2f} max {d3[m].max():7.2f}")
    xs = cst[buf[:waves, 0] > 0][:, 12:16]
    if (xs > 0).all(axis=1).any():                         # scratch sub-stamps of T3 (an experiment build only): issue | plan barrier | record landed | plane evaluation | rest
        okx = (xs > 0).all(axis=1) & (W == 0)
        raw = buf[:waves][buf[:waves, 0] > 0].astype(np.int64)      # (st is normalised per XCD; the sub-stamps are not)
        seg = np.stack([xs[:, 0] - raw[:, 2], xs[:, 1] - xs[:, 0], xs[:, 2] - xs[:, 1], xs[:, 3] - xs[:, 2], raw[:, 3] - xs[:, 3]], axis=1)[okx] / 2100.0
        print("  T3 of waves without candidate pairs, us @2.1 GHz: " + ", ".join(f"{n} mean {seg[:, k].mean():.2f} p95 {np.percentile(seg[:, k], 95):.2f}" for k, n in
              enumerate(["key/hash/issue", "plan barrier", "record lands", "plane evaluation", "rest of T3"])))
    slow = np.argsort(-d3)[:8]
    print('  slowest T3 waves (index, W, T3):', [(int(k), int(W[k]), round(float(d3[k]), 1)) for k in slow])
    live = buf[:waves, 0] > 0
    r0 = rt[live, 0].min()
    start = (rt[live, 0] - r0) * 0.01; fin = (rt[live, 1] - r0) * 0.01
    print(f"chip-wide clock (us): wave start p5 {np.percentile(start, 5):.2f} p50 {np.percentile(start, 50):.2f} p95 {np.percentile(start, 95):.2f} max {start.max():.2f}; "
          f"wave end p5 {np.percentile(fin, 5):.2f} p50 {np.percentile(fin, 50):.2f} p95 {np.percentile(fin, 95):.2f} max {fin.max():.2f}; "
          f"wave life p50 {np.percentile(fin - start, 50):.2f} max {(fin - start).max():.2f}")
    print(f"solve kernel: starts {(sol_rt[0] - r0) * 0.01:.2f} us after the residual kernel's first wave ({(sol_rt[0] - rt[live, 1].max()) * 0.01:.2f} after its last), runs {(sol_rt[1] - sol_rt[0]) * 0.01:.2f}")
    blk = np.arange(waves)[live] // 4
    late = np.argsort(-fin)[:8]
    print("  last waves to finish (wave, block, start, end, W):", [(int(np.arange(waves)[live][k]), int(blk[k]), round(float(start[k]), 2), round(float(fin[k]), 2), int(st[k, 7])) for k in late])
    print("  their phases T1..T6 (us):", [[round(float(st[k, j] - st[k, j - 1]) / 2100.0, 2) for j in range(1, 7)] for k in late[::2]])
    for k in late[::4]:
        w = int(np.arange(waves)[live][k])
        c = cst[w]
        base0 = int(buf[w, 2])
        print(f"  wave {w}: stamps after T2 end (us): round0 {[round((int(x) - base0) / 2100.0, 2) if x else None for x in c[:6]]} round1 {[round((int(x) - base0) / 2100.0, 2) if x else None for x in c[6:12]]}; T3 end {(int(buf[w, 3]) - base0) / 2100.0:.2f}")
    end = (st[:, 6] - t0) * tick_us
    print(f"wave end time: mean {end.mean():.2f} p50 {np.percentile(end, 50):.2f} p95 {np.percentile(end, 95):.2f} max {end.max():.2f} us")


if __name__ == "__main__":
    main()
