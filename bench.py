#!/usr/bin/env python
"""bench.py — ESIKF measurement-update throughput on MI355X (driver contract: one JSON line from rank 0).

Workload (BASELINE.json configs[1], "C2"): 100 000 synthetic LiDAR points, point-to-plane residual + Jacobian + H/b
reduction + 19-dim solve.  One STEP = one complete ESIKF iteration over the whole scan (k_lidar_residual + k_lidar_solve),
inputs (scan, body covariances, VoxelMap snapshot, states) resident in HBM.  metric = residual+Jacobian evaluations / s =
points x steps / wall time (whole job, all ranks).  For --gpus N every rank runs the same-sized independent frame
(frames shard embarrassingly; no data-path collective): scaling = "weak".

Extra objects on the same line:
  roofline      achieved = 276 B (SURVEY §8d algorithmic bytes per LiDAR point-iteration) x points / average
                k_lidar_residual duration measured with HIP events on the launching stream; peak = 8 TB/s HBM3E.
  cpu_baseline  the oracle ("port": restated reference CPU path, -O3 -march=native -fopenmp, 4 threads = the reference's
                MP_PROC_NUM cap) timed on this host on the same scan: full StateEstimation calls, evals = points x iterations.
  extra         visual path (C3: 2k patches) and full 5-iteration updates (C4-style), informational.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

LIDAR_BYTES_PER_EVAL = 276.0     # 12 (xyz f32) + 32 (hash probe) + 232 (plane record f64)        SURVEY.md §8(d)
VISUAL_BYTES_PER_PATCH = 413.0   # 121 (u8 window) + 256 (ref patch f32) + 36 (pos, level, expo)  SURVEY.md §8(d)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def make_states(livo2, sc):
    cur = livo2.State.from_pose(sc.R_prior, sc.t_prior, sc.P, inv_expo=getattr(sc, "tau_prior", 1.0))
    return cur, cur.copy()


PLANE_FIT_BYTES_PER_POINT = 96.0        # point_w 24 + var 72 (each point read once per pass; second pass is the same lines)


def plane_fit_groups(n_groups=20000, seed=4):
    """voxel point groups of the size UpdateVoxelMap re-fits (6..60 points, update_size_threshold_ 5 .. max_points_num_ 50)"""
    rng = np.random.default_rng(seed)
    cnt = rng.integers(6, 61, n_groups)
    off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    N = int(off[-1])
    gid = np.repeat(np.arange(n_groups), cnt)
    Q = np.linalg.qr(rng.normal(size=(n_groups, 3, 3)))[0]
    ext = np.where((np.arange(n_groups) % 5 == 4)[:, None], [0.12, 0.11, 0.1], [0.15, 0.12, 0.01])       # 80 % planar patches, 20 % blobs
    local = rng.normal(size=(N, 3)) * ext[gid]
    pw = (np.einsum("nij,nj->ni", Q[gid], local) + rng.uniform(-40, 40, (n_groups, 3))[gid]).astype(np.float32).astype(np.float64)
    A = rng.normal(size=(N, 3, 3))
    var = 1e-4 * (A @ A.transpose(0, 2, 1) + 0.1 * np.eye(3))
    return pw, var.reshape(N, 9), off


def cpu_baseline(sc, budget_s=20.0):
    """Oracle (restated reference CPU path) timed on this host.  Bounded: at most `budget_s` seconds of StateEstimation calls."""
    from oracle import orc
    from tests import helpers as H
    kind = "port"
    try:
        path = orc.build("fast", out_dir=tempfile.mkdtemp(prefix="orc_fast_"))     # -march=native: must be compiled on this host
        flags = "-O3 -march=native -funroll-loops -fopenmp"
    except Exception:
        path = None
        flags = "-O2 -ffp-contract=off -fopenmp (golden build; fast build failed)"
    lib = orc.load(path)
    om = orc.OracleMap.from_flat(sc.fmap, lib)
    cur, prop = H.states(sc, orc.StatePOD)
    out = {}
    ncores = os.cpu_count() or 1
    for threads in (4, 1, ncores):
        cfg = orc.lidar_cfg(sc.cfg, sc.extR, sc.extT, num_threads=threads)
        orc.lidar_state_estimation(om, cfg, sc.xyz, cur, prop, want_points=False)      # warm-up
        secs, evals, runs = 0.0, 0, 0
        while secs < budget_s * (0.6 if threads == 4 else 0.2) and runs < 40:
            r = orc.lidar_state_estimation(om, cfg, sc.xyz, cur, prop, want_points=False)
            secs += r["seconds"]; evals += len(sc.xyz) * r["n_iters"]; runs += 1
        out[threads] = (evals / secs, runs)
    pw, var, off = plane_fit_groups()
    orc.init_plane_batch(pw, var, off, 0.0025, lib)
    _, fit_s = orc.init_plane_batch(pw, var, off, 0.0025, lib)
    from scenarios import synth as _synth
    rs = _synth.retrieve_scenario(seed=21, n_cand=2000)
    orc.warp_candidates(rs, lib)
    warp_s = min(orc.warp_candidates(rs, lib)["seconds"] for _ in range(3))
    from tests import imu_inputs as IMU
    ist = IMU.make_state(orc, orc.StatePOD, 0)
    orc.imu_propagate(ist, IMU.make_steps(0, n=20), IMU.CFG, lib)
    imu_us = 1e6 * min(orc.imu_propagate(ist, IMU.make_steps(0, n=20), IMU.CFG, lib)[2] for _ in range(5))
    ss = _synth.select_scenario(seed=71, n_pg=10000, n_vis=100000)
    orc.visual_select(ss, lib)
    sel_s = min(orc.visual_select(ss, lib)["seconds"] for _ in range(3))
    cs = _synth.retrieve_chain_scenario(seed=81, n_pg=10000, n_vis=30000, grid_n_height=102, normal_en=True)
    tch = []
    for _ in range(3):
        t0 = time.perf_counter(); orc.visual_retrieve(cs, lib); tch.append(time.perf_counter() - t0)
    raw = _synth.raw_scan_scenario(seed=51, n_raw=240000)
    tpre = []
    for _ in range(3):
        t0 = time.perf_counter()
        u_ = orc.undistort(raw.xyz, raw.curvature, raw.poses, raw.rot_end, raw.pos_end, raw.extR, raw.extT, lib)
        orc.voxel_grid(u_, raw.leaf, lib)
        tpre.append(time.perf_counter() - t0)
    return {"imu_propagate_us_20_samples": imu_us, "select_seconds_1thread": sel_s, "retrieve_from_map_seconds_1thread": min(tch), "preprocess_points_per_s_1thread": len(raw.xyz) / min(tpre), "plane_fit_points_per_s_1thread": len(pw) / fit_s, "plane_fit_groups": len(off) - 1, "retrieve_candidates_per_s_1thread": len(rs.pos) / warp_s,
            "value": out[4][0], "unit": "evals/s", "cores": 4, "kind": kind,
            "sample": f"{out[4][1]} full StateEstimation calls (5 iterations each) on the same {len(sc.xyz)}-point scan, OpenMP 4 threads (reference MP_PROC_NUM cap), {flags}",
            "value_1thread": out[1][0], "value_all_cores": out[ncores][0], "host_cores": ncores}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--points", type=int, default=100000)
    ap.add_argument("--scan-order", choices=["voxelgrid", "random"], default="voxelgrid",
                    help="voxelgrid: scan passed through the 0.1 m centroid voxel-grid filter like feats_down_body (LIVMapper.cpp:351-352), "
                         "points ordered by leaf index; random: raw random-ray order (no spatial coherence)")
    ap.add_argument("--batch", type=int, default=16, help="frames per launch in the batched-frames leg (extra.batched); 0 disables it")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the informational visual / full-update legs")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the ESIKF path has no CPU fallback)")
    torch.cuda.set_device(local_rank)

    from scenarios import synth
    from tests import helpers as H
    livo2 = importlib.import_module("fast-livo2_amd")

    # ---- workload: C2 ----------------------------------------------------------------------------------------------
    sc = synth.lidar_scenario(seed=2 + rank, n_points=args.points, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays_factor=12,
                              downsample=(synth.AVIA["filter_size_surf"] if args.scan_order == "voxelgrid" else None))
    n = len(sc.xyz)
    ctx = livo2.Context(local_rank)
    cfg = H.lidar_cfg_product(sc)
    ctx.upload_map(sc.fmap)
    ctx.set_scan(sc.xyz, cfg)
    cur, prop = make_states(livo2, sc)

    def barrier():
        ctx.synchronize(); torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ctx.synchronize(); torch.cuda.synchronize()

    if args.warmup > 0:
        ctx.lidar_iterations_async(cur, prop, cfg, args.warmup)
    barrier()
    t0 = time.perf_counter()
    ctx.lidar_iterations_async(cur, prop, cfg, args.steps)
    ctx.synchronize(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    frames = importlib.import_module("fast-livo2_amd.frames")
    elapsed = frames.max_over_ranks(elapsed, dist, device="cuda")
    n_all = int(frames.gather_results(np.array([[float(n)]]), world, dist, device="cuda").sum()) if dist is not None else n
    value = n_all * args.steps / elapsed

    # ---- frames/s leg (BASELINE "frames/sec at 1/2/4/8 GPU"): every rank replays F whole frames = scan upload (H2D, Morton sort, body
    # covariance) + full StateEstimation loop on the GPU + result read-back; frames shard round-robin, results are gathered ----
    F_per_rank = 16
    ctx.set_scan(sc.xyz, cfg); ctx.lidar_update_async(cur, prop, cfg); ctx.lidar_update_fetch()
    barrier()
    tf0 = time.perf_counter()
    local = []
    for _ in range(F_per_rank):
        ctx.set_scan(sc.xyz, cfg)
        ctx.lidar_update_async(cur, prop, cfg)
        rf = ctx.lidar_update_fetch()
        local.append(np.concatenate([np.array(rf.state.pos), [rf.n_iters, rf.iter_sums[rf.n_iters - 1].n_eff]]))
    ctx.synchronize()
    tframes = frames.max_over_ranks(time.perf_counter() - tf0, dist, device="cuda")
    gathered = frames.gather_results(np.array(local), F_per_rank * world, dist, device="cuda")
    frames_per_s = len(gathered) / tframes

    # same frames through TWO contexts (two host threads, two streams on this GPU; ctypes releases the GIL inside the calls): the H2D + sort of
    # one frame overlaps the update of another
    import threading
    ctx_b = livo2.Context(local_rank)
    ctx_b.upload_map(sc.fmap)
    def replay(c, k):
        for _ in range(k):
            c.set_scan(sc.xyz, cfg); c.lidar_update_async(cur, prop, cfg); c.lidar_update_fetch()
    replay(ctx_b, 1)
    barrier(); ctx_b.synchronize()
    best2 = 0.0
    for _ in range(3):                                    # host-thread scheduling makes single runs noisy: best of three
        tf0 = time.perf_counter()
        th = [threading.Thread(target=replay, args=(c, F_per_rank // 2)) for c in (ctx, ctx_b)]
        [t.start() for t in th]; [t.join() for t in th]
        best2 = max(best2, (F_per_rank // 2) * 2 * world / frames.max_over_ranks(time.perf_counter() - tf0, dist, device="cuda"))
    frames_per_s_2ctx = best2
    ctx_b.close()

    # ---- roofline leg: HIP-event duration of the dominant kernel over the same launch sequence ----------------------------
    ctx.kernel_timing(True)
    ctx.kernel_timing_read(0)
    ctx.kernel_timing_read(2)
    ctx.lidar_iterations_async(cur, prop, cfg, args.steps)
    ms_res, n_res = ctx.kernel_timing_read(0)
    ms_sol, n_sol = ctx.kernel_timing_read(2)
    ctx.kernel_timing(False)
    res_us = 1e3 * ms_res / max(n_res, 1)
    sol_us = 1e3 * ms_sol / max(n_sol, 1)
    achieved = LIDAR_BYTES_PER_EVAL * n / (res_us * 1e-6) / 1e9
    # HBM traffic per launch cannot be read from inside this process: it comes from the separate rocprofv3 --pmc passes recorded in
    # profiles/ (only reported when they were taken on this very workload)
    traffic, traffic_note = None, "no PMC pass recorded for this workload; see profiles/"
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if rec["points"] == n and rec["scan_order"] == args.scan_order:
            traffic = (rec["fetch_size_kb_reported"] * rec["fetch_correction"] + rec["write_size_kb"]) * 1024.0
            traffic_note = rec["source"]
    except Exception:
        pass
    # achievable HBM bandwidth on this box: a device-to-device copy of 1 GiB (read + write counted), SURVEY 8d
    src_t = torch.empty(1 << 28, dtype=torch.float32, device="cuda"); dst_t = torch.empty_like(src_t)
    dst_t.copy_(src_t); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        dst_t.copy_(src_t)
    ev1.record(); torch.cuda.synchronize()
    copy_gbs = 5 * 2 * src_t.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del src_t, dst_t
    roofline = {"copy_kernel_GBps": copy_gbs, "frac_of_copy_kernel": achieved / copy_gbs, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": "k_lidar_residual", "kernel_us": res_us, "bytes_per_launch": LIDAR_BYTES_PER_EVAL * n, "solve_kernel_us": sol_us,
                "timing": "HIP event pair around every launch on the launching stream (includes the dependent-launch gap; rocprofv3 kernel-only average in profiles/)",
                "traffic_unit": "bytes/launch", "traffic_note": traffic_note}

    extra = {"frames_per_s": frames_per_s, "frames_per_s_two_contexts": frames_per_s_2ctx, "frame_points": n, "frames": int(F_per_rank * world),
             "frame_def": "set_scan (H2D + Morton sort + body cov) + full StateEstimation loop + result read-back"}
    if rank == 0 and not args.no_extra:
        try:
            # full StateEstimation (<=5 iterations with convergence logic), end-to-end incl. result read-back
            reps = 20
            ctx.lidar_update_async(cur, prop, cfg); r0 = ctx.lidar_update_fetch()
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.lidar_update_async(cur, prop, cfg); r0 = ctx.lidar_update_fetch()
            dt = (time.perf_counter() - t1) / reps
            extra["lidar_full_update_ms"] = dt * 1e3
            extra["lidar_full_update_iters"] = int(r0.n_iters)
            extra["lidar_n_eff"] = int(r0.iter_sums[r0.n_iters - 1].n_eff)
            # visual C3: 2k patches
            vs = synth.visual_scenario(seed=3, n_patches=2000)
            vcfg = H.visual_cfg_product(vs)
            vcur, vprop = make_states(livo2, vs)
            ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
            ctx.visual_iterations_async(0, vcur, vprop, vcfg, args.warmup or 1); ctx.synchronize()
            t1 = time.perf_counter()
            ctx.visual_iterations_async(0, vcur, vprop, vcfg, args.steps); ctx.synchronize()
            dtv = time.perf_counter() - t1
            extra["visual_evals_per_s"] = 64.0 * len(vs.pos) * args.steps / dtv
            extra["visual_patches"] = len(vs.pos)
            ctx.kernel_timing(True); ctx.kernel_timing_read(1)
            ctx.visual_iterations_async(0, vcur, vprop, vcfg, args.steps)
            ms_v, n_v = ctx.kernel_timing_read(1)
            ctx.kernel_timing(False)
            v_us = 1e3 * ms_v / max(n_v, 1)
            extra["visual_kernel_us"] = v_us
            extra["visual_achieved_GBps"] = VISUAL_BYTES_PER_PATCH * len(vs.pos) / (v_us * 1e-6) / 1e9
            reps = 20
            ctx.visual_update_async(vcur, vprop, vcfg); ctx.visual_update_fetch()
            t1 = time.perf_counter()
            for _ in range(reps):
                ctx.visual_update_async(vcur, vprop, vcfg); rv = ctx.visual_update_fetch()
            extra["visual_full_update_ms"] = (time.perf_counter() - t1) / reps * 1e3
            extra["visual_full_update_steps"] = int(rv.n_steps)
            # C3 (BASELINE configs[2]): the LiDAR iteration of the 100k-point scan and the visual iteration of the 2k patches in flight together
            # (two contexts = two streams on this GPU; the two updates of a frame are separate ESIKF updates in the reference, LIVMapper.cpp:370 / vio.cpp:1810)
            ctx_v = livo2.Context(local_rank)
            ctx_v.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
            ctx.lidar_iterations_async(cur, prop, cfg, 5); ctx_v.visual_iterations_async(0, vcur, vprop, vcfg, 5); ctx.synchronize(); ctx_v.synchronize()
            t1 = time.perf_counter()
            ctx.lidar_iterations_async(cur, prop, cfg, args.steps); ctx_v.visual_iterations_async(0, vcur, vprop, vcfg, args.steps)
            ctx.synchronize(); ctx_v.synchronize()
            dtc = time.perf_counter() - t1
            extra["c3_lidar_plus_visual"] = {"evals_per_s": (n + 64.0 * len(vs.pos)) * args.steps / dtc, "lidar_points": n, "visual_patches": len(vs.pos),
                                             "ms_per_step": 1e3 * dtc / args.steps, "note": "one LiDAR ESIKF iteration and one visual iteration (level 0) per step on two streams"}
            ctx_v.close()
            # One Avia-sized frame through every device stage built so far (C1 sizes: 24 000 raw points / scan, a few hundred patches): the
            # reference's budget for this is 100 ms per frame on <= 4 host threads
            sc1 = synth.lidar_scenario(seed=1, n_points=10000, downsample=0.1)
            raw1 = synth.raw_scan_scenario(seed=1, n_raw=24000)
            cfg1 = H.lidar_cfg_product(sc1)
            cur1, prop1 = make_states(livo2, sc1)
            rs1 = synth.retrieve_scenario(seed=2, n_cand=400)
            vs1 = synth.visual_scenario(seed=3, n_patches=8); vs1.img, vs1.cam = rs1.img, rs1.cam
            vcfg1 = H.visual_cfg_product(vs1)
            vcur1, vprop1 = make_states(livo2, vs1)
            fpw1, fvar1, foff1 = plane_fit_groups(n_groups=800, seed=5)
            ctx.upload_map(sc1.fmap)
            stage = {}
            for rep in range(6):
                t = [time.perf_counter()]
                nd1 = ctx.preprocess_scan(raw1.xyz, raw1.curvature, raw1.poses, raw1.rot_end, raw1.pos_end, raw1.leaf, cfg1, want=False)[0]; t.append(time.perf_counter())
                ctx.set_scan(sc1.xyz, cfg1)                                   # (the synthetic raw scan is not registered to this map: update the matching scan)
                t.append(time.perf_counter())
                ctx.lidar_update(cur1, prop1, cfg1); t.append(time.perf_counter())
                ctx.plane_fit_batch(fpw1, fvar1, foff1, 0.0025); t.append(time.perf_counter())
                ctx.retrieve_warp(rs1, want_patches=False); t.append(time.perf_counter())
                ctx.visual_update(vcur1, vprop1, vcfg1); t.append(time.perf_counter())
                if rep:
                    for name, a, b in (("preprocess_scan", 0, 1), ("lidar_update", 2, 3), ("plane_fit_800_voxels", 3, 4), ("retrieve_warp_400", 4, 5), ("visual_update", 5, 6)):
                        stage.setdefault(name, []).append((t[b] - t[a]) * 1e3)
            extra["avia_frame_stages_ms"] = {k: float(np.median(v)) for k, v in stage.items()}
            extra["avia_frame_stages_ms"]["sum"] = float(sum(np.median(v) for v in stage.values()))
            extra["avia_frame_stages_ms"]["note"] = "host-synchronous calls through the Python wrappers incl. H2D/D2H: 24 000 raw points -> %d, 10 000-point LiDAR update, 800 voxel re-fits, 400 retrieval candidates, visual update on the survivors" % nd1
            # The LiDAR-inertial part of a frame as ONE call (livo2_lio_frame: IMU propagation -> undistortion + voxel grid -> StateEstimation) next to the same
            # three stages called one after the other, on a raw scan that is registered to its map
            try:
                lf = synth.lio_frame_scenario(seed=61, n_raw=24000, n_steps=20)
                lcfg = H.lidar_cfg_product(lf.sc)
                lst = livo2.State.from_pose(lf.sc.R_prior, lf.sc.t_prior, lf.sc.P)
                lst.inv_expo = lf.inv_expo; lst.vel[:] = lf.vel.tolist(); lst.bg[:] = lf.bg.tolist(); lst.ba[:] = lf.ba.tolist(); lst.grav[:] = lf.grav.tolist()
                licfg = livo2.ImuCfg()
                for k in ("cov_gyr", "cov_acc", "cov_bias_gyr", "cov_bias_acc"):
                    getattr(licfg, k)[:] = lf.imu[k]
                licfg.cov_inv_expo, licfg.G_m_s2, licfg.mean_acc_norm = lf.imu["cov_inv_expo"], lf.imu["G_m_s2"], lf.imu["mean_acc_norm"]
                licfg.ba_bg_est_en = licfg.gravity_est_en = licfg.exposure_estimate_en = 1
                ctx.upload_map(lf.sc.fmap)
                t_seq, t_one = [], []
                for rep in range(6):
                    t0 = time.perf_counter()
                    lprop, lposes = ctx.imu_propagate(lst, lf.steps, licfg)
                    ctx.preprocess_scan(lf.sc.xyz, lf.curvature, np.vstack([lf.first_pose, lposes]), np.array(lprop.rot).reshape(3, 3), np.array(lprop.pos), synth.AVIA["filter_size_surf"], lcfg, want=False)
                    lres_seq, _ = ctx.lidar_update(lprop, lprop, lcfg)
                    t1 = time.perf_counter()
                    lres, lnd, _, _ = ctx.lio_frame(lst, lf.steps, licfg, lf.first_pose, lf.sc.xyz, lf.curvature, synth.AVIA["filter_size_surf"], lcfg, want_poses=False)
                    t2 = time.perf_counter()
                    if rep:
                        t_seq.append((t1 - t0) * 1e3); t_one.append((t2 - t1) * 1e3)
                extra["lio_frame"] = {"raw_points": len(lf.sc.xyz), "feats_down_size": int(lnd), "imu_steps": len(lf.steps), "iterations": int(lres.n_iters),
                                      "one_call_ms": float(np.median(t_one)), "three_calls_ms": float(np.median(t_seq)), "same_result": bytes(lres.state) == bytes(lres_seq.state),
                                      "note": "livo2_lio_frame vs livo2_imu_propagate + livo2_lidar_preprocess_scan + livo2_lidar_update through the Python wrappers, host-synchronous, "
                                              "incl. H2D of the raw scan and D2H of the result; state_propagat, IMUpose and feats_down_body stay on the device in the one-call form"}
            except Exception as exc:                                   # informational leg: never take the bench line down with it
                extra["lio_frame"] = {"error": repr(exc)}
            ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)
            # IMU forward propagation (20 samples = 100 ms at 200 Hz)
            from tests import imu_inputs as IMU
            ist = livo2.State.from_pose(sc.R_prior, sc.t_prior, sc.P); ist.grav[:] = [0.0, 0.0, -9.81]
            icfg = livo2.ImuCfg()
            for k in ("cov_gyr", "cov_acc", "cov_bias_gyr", "cov_bias_acc"):
                getattr(icfg, k)[:] = IMU.CFG[k]
            icfg.cov_inv_expo, icfg.G_m_s2, icfg.mean_acc_norm = IMU.CFG["cov_inv_expo"], IMU.CFG["G_m_s2"], IMU.CFG["mean_acc_norm"]
            icfg.ba_bg_est_en = icfg.gravity_est_en = icfg.exposure_estimate_en = 1
            isteps = IMU.make_steps(0, n=20)
            ctx.imu_propagate(ist, isteps, icfg)
            us = []
            for _ in range(5):
                ctx.imu_propagate(ist, isteps, icfg); us.append(ctx.imu_last_kernel_us())
            extra["imu_propagate"] = {"samples": 20, "kernel_us": float(np.median(us)), "us_per_sample": float(np.median(us)) / 20,
                                      "note": "k_imu_propagate: one block, sequential over the samples (19x19 F P F^T + Q per sample); latency-bound, on par with a host core "
                                              "(cpu_baseline.imu_propagate_us_20_samples) — built so that state_propagat / IMUpose can be produced next to their consumers"}
            # raw scan -> UndistortPcl -> pcl::VoxelGrid -> resident scan
            raw = synth.raw_scan_scenario(seed=51, n_raw=240000)
            ctx.preprocess_scan(raw.xyz, raw.curvature, raw.poses, raw.rot_end, raw.pos_end, raw.leaf, cfg, want=False)
            us, t1 = [], time.perf_counter()
            for _ in range(5):
                nd, _, _ = ctx.preprocess_scan(raw.xyz, raw.curvature, raw.poses, raw.rot_end, raw.pos_end, raw.leaf, cfg, want=False); us.append(ctx.preprocess_last_kernel_us())
            t_e2e = (time.perf_counter() - t1) / 5
            k_us = float(np.median(us))
            extra["preprocess_scan"] = {"raw_points": len(raw.xyz), "feats_down_size": nd, "imu_poses": len(raw.poses), "kernel_us": k_us,
                                        "points_per_s_kernel": len(raw.xyz) / (k_us * 1e-6), "achieved_GBps": 44.0 * len(raw.xyz) / (k_us * 1e-6) / 1e9,
                                        "points_per_s_with_h2d_and_scan_setup": len(raw.xyz) / t_e2e,
                                        "note": "k_undistort + voxel grid (min/max, keys, rocPRIM radix sort, heads, scan, centroids); 44 B/point = xyz+time read, xyz written, "
                                                "xyz re-read, centroid share; the event span covers ~20 small launches (rocprofv3: ~115 us of kernel time at 240k points, the rest is enqueue gaps); "
                                                "CPU figure in cpu_baseline.preprocess_points_per_s_1thread"}
            ctx.set_scan(sc.xyz, cfg)
            # selection half of retrieveFromVisualSparseMap (scan voxels + depth image, nearest visual point per grid cell, depth continuity)
            ss = synth.select_scenario(seed=71, n_pg=10000, n_vis=100000)
            ctx.visual_map_upload(ss.pos, ss.keys, ss.active)
            ctx.visual_select(ss)
            us, t1 = [], time.perf_counter()
            for _ in range(5):
                so = ctx.visual_select(ss); us.append(ctx.select_last_kernel_us())
            t_e2e = (time.perf_counter() - t1) / 5
            k_us = float(np.median(us))
            extra["visual_select"] = {"scan_points": len(ss.pg), "visual_map_points": len(ss.pos), "cells_selected": int((so["cell_point"] >= 0).sum()), "kernel_us": k_us,
                                      "visual_points_per_s_kernel": len(ss.pos) / (k_us * 1e-6), "calls_per_s_with_h2d_d2h": 1.0 / t_e2e,
                                      "note": "memsets + k_sel_scan + k_sel_points + k_sel_cells (vio.cpp:385-486, 598-635) with the visual map resident; CPU figure in cpu_baseline.select_seconds_1thread"}
            # per-point tail of retrieveFromVisualSparseMap (warp matrix, search level, warpAffine x L, getImagePatch, gates, compaction)
            rs = synth.retrieve_scenario(seed=21, n_cand=2000)
            ctx.retrieve_warp(rs, want_patches=False)
            us, t1 = [], time.perf_counter()
            for _ in range(5):
                ro = ctx.retrieve_warp(rs, want_patches=False); us.append(ctx.retrieve_last_kernel_us())
            t_e2e = (time.perf_counter() - t1) / 5
            k_us = float(np.median(us))
            Lr = int(rs.cfg["patch_pyrimid_level"])
            bytes_per_cand = 200.0 + 81.0 * (Lr + 1) + 256.0 * Lr        # descriptors + (L reference windows + current window, u8) + warped patches written
            extra["retrieve_warp"] = {"candidates": len(rs.pos), "accepted": ro["n_accepted"], "levels": Lr, "kernel_us": k_us,
                                      "candidates_per_s_kernel": len(rs.pos) / (k_us * 1e-6), "bytes_per_candidate": bytes_per_cand,
                                      "achieved_GBps": bytes_per_cand * len(rs.pos) / (k_us * 1e-6) / 1e9, "candidates_per_s_with_h2d_d2h": len(rs.pos) / t_e2e,
                                      "note": "k_warp_candidates + k_warp_scan + k_warp_gather (vio.cpp:698-767); CPU figure in cpu_baseline.retrieve_candidates_per_s_1thread"}
            # the whole retrieveFromVisualSparseMap as one chain (selection -> reference-patch choice -> tail), map + observations resident
            cs = synth.retrieve_chain_scenario(seed=81, n_pg=10000, n_vis=30000, grid_n_height=102, normal_en=True)     # grid_size 5 as in config/avia.yaml
            ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
            ctx.visual_obs_upload(cs)
            ctx.visual_retrieve_from_map(cs, want_patches=False)
            us, t1 = [], time.perf_counter()
            for _ in range(5):
                ctx.visual_obs_upload(cs)                      # resets ref_patch: every call makes the first-time choices again
                co = ctx.visual_retrieve_from_map(cs, want_patches=False); us.append(ctx.retrieve_from_map_last_kernel_us())
            t_e2e = (time.perf_counter() - t1) / 5
            k_us = float(np.median(us))
            extra["retrieve_from_map"] = {"scan_points": len(cs.sel.pg), "visual_map_points": len(cs.sel.pos), "observations": int(cs.obs_offset[-1]),
                                          "grid_cells": int(cs.sel.grid_n_width * cs.sel.grid_n_height), "candidates": co["n_candidates"], "accepted": co["n_accepted"],
                                          "kernel_us": k_us, "calls_per_s_with_obs_upload_h2d_d2h": 1.0 / t_e2e,
                                          "note": "selection + k_choose_ref + scan + k_gather_candidates + tail in one chain of launches (vio.cpp:352-780), no host round trip; "
                                                  "CPU figure in cpu_baseline.retrieve_from_map_seconds_1thread"}
            # batched init_plane (plane fit + plane covariance) on the device
            fpw, fvar, foff = plane_fit_groups()
            ctx.plane_fit_batch(fpw, fvar, foff, 0.0025)
            us = []
            t1 = time.perf_counter()
            for _ in range(5):
                fo = ctx.plane_fit_batch(fpw, fvar, foff, 0.0025); us.append(ctx.plane_fit_last_kernel_us())
            t_e2e = (time.perf_counter() - t1) / 5
            k_us = float(np.median(us))
            extra["plane_fit"] = {"groups": len(foff) - 1, "points": len(fpw), "planes": int(sum(o.is_plane for o in fo)), "kernel_us": k_us,
                                  "points_per_s_kernel": len(fpw) / (k_us * 1e-6), "achieved_GBps": PLANE_FIT_BYTES_PER_POINT * len(fpw) / (k_us * 1e-6) / 1e9,
                                  "frac_of_hbm_peak": PLANE_FIT_BYTES_PER_POINT * len(fpw) / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                  "points_per_s_with_h2d_d2h": len(fpw) / t_e2e,
                                  "note": "k_plane_fit: 8 lanes per voxel group (64 for groups > 64 points); VoxelOctoTree::init_plane (voxel_map.cpp:55-135); CPU figure in cpu_baseline.plane_fit_points_per_s_1thread"}
        except Exception as exc:                                   # informational legs must never take the bench line down with them
            extra["error"] = repr(exc)
            try:
                ctx.upload_map(sc.fmap); ctx.set_scan(sc.xyz, cfg)  # the batched leg below expects the C2 map and scan resident
            except Exception:
                pass


    # ---- batched frames (BASELINE configs[4] shape, extra only): B scans of the C2 size against the resident map, one residual grid +
    # one solve block per frame per ESIKF iteration.  Frames differ (own 97 % subset of the scan, own prior perturbation). ----
    if args.batch > 0:
        B = args.batch
        rngb = np.random.default_rng(100 + rank)
        scans, bst = [], []
        for f in range(B):
            keep = np.sort(rngb.permutation(n)[: int(0.97 * n)])
            scans.append(sc.xyz[keep])
            Rf = sc.R_prior @ synth.so3_exp(rngb.normal(0, np.deg2rad(0.2), 3))
            bst.append(livo2.State.from_pose(Rf, sc.t_prior + rngb.normal(0, 0.02, 3), sc.P))
        ctx.batch_set_scans(scans, cfg)
        npts = int(sum(len(x) for x in scans))
        ctx.batch_iterations_async(bst, bst, cfg, max(args.warmup, 1)); barrier()
        tb0 = time.perf_counter()
        ctx.batch_iterations_async(bst, bst, cfg, args.steps)
        ctx.synchronize()
        tb = frames.max_over_ranks(time.perf_counter() - tb0, dist, device="cuda")
        ctx.kernel_timing(True); ctx.kernel_timing_read(0); ctx.kernel_timing_read(2)
        ctx.batch_iterations_async(bst, bst, cfg, args.steps)
        bms_res, bn_res = ctx.kernel_timing_read(0)
        bms_sol, bn_sol = ctx.kernel_timing_read(2)
        ctx.kernel_timing(False)
        bres_us = 1e3 * bms_res / max(bn_res, 1)
        # whole frames: upload + sort + full update + read-back, B at a time
        ctx.batch_set_scans(scans, cfg); ctx.batch_update_async(bst, bst, cfg); ctx.batch_update_fetch()
        barrier()
        reps_b = 4
        tb1 = time.perf_counter()
        for _ in range(reps_b):
            ctx.batch_set_scans(scans, cfg); ctx.batch_update_async(bst, bst, cfg); rb = ctx.batch_update_fetch()
        tfb = frames.max_over_ranks(time.perf_counter() - tb1, dist, device="cuda")
        bach = LIDAR_BYTES_PER_EVAL * npts / (bres_us * 1e-6) / 1e9
        extra["batched"] = {"frames_per_launch": B, "points_per_launch": npts, "evals_per_s": world * npts * args.steps / tb, "ms_per_step": 1e3 * tb / args.steps,
                            "residual_kernel_us": bres_us, "solve_kernel_us": 1e3 * bms_sol / max(bn_sol, 1),
                            "roofline": {"bound": "hbm", "achieved": bach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bach / HBM_PEAK_GBS, "frac_of_copy_kernel": bach / copy_gbs,
                                         "kernel": "k_lidar_residual_batch", "bytes_per_launch": LIDAR_BYTES_PER_EVAL * npts},
                            "frames_per_s": world * B * reps_b / tfb, "full_update_iters": [int(r.n_iters) for r in rb],
                            "note": "same device code as the single-scan path with 64-point blocks, B independent (scan, state) problems per grid; same decisions as B single calls, sums equal to rounding (tests/test_batch_gpu.py)"}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(sc)

    if rank == 0:
        line = {
            "metric": "residual+Jacobian evals/sec (LiDAR+visual) per ESIKF iter", "value": value, "unit": "evals/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C2: 100k synthetic LiDAR points, point-to-plane residual+Jacobian+H/b+solve, 1 ESIKF iteration per step (BASELINE.json configs[1])",
                       "points_per_gpu": n, "scan_order": args.scan_order, "plane_records": int(sc.fmap.n_planes), "voxels": int(len(sc.fmap.root_node)), "parallelism": f"frames x{world} (no collective on the data path)"},
            "roofline": roofline, "cpu_baseline": cpu, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
