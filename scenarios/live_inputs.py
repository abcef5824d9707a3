"""Inputs of fast-livo2_amd/host/live_chain (chained live LIO + VIO frames through the C++ shim) — ONE scene: a room, a sensor path through it, per frame k
  * a down-sampled LiDAR scan taken at the true pose T_k and the commanded motion from T_{k-1} (stand-in for the IMU propagation),
  * the visual map as the (out-of-scope) map maintenance of the reference would hold it when frame k arrives: visual points ON THE SURFACES OF THE SAME ROOM seen
    from T_k, each observed by earlier frames standing close to T_k (their patches come from the texture the current image shows), and the current image.
The runner chains what the reference chains (LIVMapper.cpp:135-136, 256-257, 371, 413-426; vio.cpp:1799-1810): StateEstimation from the propagated state, map
update from the LIO posterior (the next frame reads THAT map), new_frame_->T_f_w_ and the scan's world points `pg` from the LIO posterior, visual update on the
shared state starting at the LIO posterior, next frame propagated from the VIO posterior.
Dump layout (binary, as fast-livo2_amd/host/shim_demo.cpp reads it): seq_* = first sweep + scans + motion + state0, vis_cfg = camera + extrinsics,
chain_cfg = [n_frames, normal_en, ncc_en, ncc_thre, outlier_threshold, L, border, grid_n_height], chain<k>_* = visual map and image of frame k.
Growing-map mode (make_live(grow=n), SIZES["avia_grow"]): ONE visual map, chain0_* = the map when frame 0 arrives, grow<k>_* = the scripted maintenance that runs
after frame k (scenarios/visual_map_growth.py: new points, new observations, per touched point pop / push-front / ref_patch action / normal flip / removal) and
grow<k>_img = the current image of frame k >= 1; grow_cfg = [number of scripts]."""
import os

import numpy as np

from scenarios import synth
from scenarios.visual_map_growth import GrowingMap, OBS_KEYS


def _state_vec(R, t, P, inv_expo=1.0):
    return np.concatenate([np.asarray(R, float).ravel(), np.asarray(t, float), [inv_expo], np.zeros(12), np.asarray(P, float).ravel()])


SIZES = {
    "avia": dict(),                                                           # 24 000 rays per scan (~12.5 k points after the 0.1 m filter), 30 000 visual points per frame
    "c4": dict(n_raw=620000, max_points=200000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays=1600000, n_vis=120000, n_frames=6),
    "test": dict(n_frames=5, n_raw=8000, map_rays=60000, n_vis=6000),       # tests/test_live_chain_gpu.py
    # ONE visual map that GROWS: frame 0's map + per frame <= 100 new points, one new observation pushed to the front of ~100 obs_ lists, ref_patch / normal changes,
    # a few deletions, one new reference image (scenarios/visual_map_growth.py) — what the incremental mirror (livo2_visual_map_apply) is for
    "avia_grow": dict(grow=100),
    "test_grow": dict(n_frames=6, n_raw=8000, map_rays=60000, n_vis=6000, grow=60),
}


def make_live(n_frames=8, n_raw=24000, map_rays=300000, room=(20.0, 20.0, 6.0), n_boxes=8, full_sphere=False, downsample=0.1, max_points=None,
              n_vis=30000, seed=191, L=2, grow=0):
    """the sequence as Python objects (deterministic in its arguments): used by write_live_dir, by oracle/live_chain.py (the same chain on the oracle) and by bench.py"""
    rng = np.random.default_rng(seed)
    c = dict(synth.AVIA["lio"])
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    scene = synth.make_room(rng, room, n_boxes)
    R0, t0 = scene.R_ws @ synth.rot_from_rpy(0.01, -0.015, 0.4), scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3
    K = n_frames
    dR = [synth.rot_from_rpy(0.0, 0.0, 0.02 * (k % 2 * 2 - 1)) for k in range(K)]
    dt = [np.array([0.06, 0.02 * (k % 3 - 1), 0.0]) for k in range(K)]
    if grow:
        # the synthetic images are textures, not renderings of the room: a visual map that lives across frames only keeps matching them while the sensor stays within
        # ~a centimetre of where its patches were taken — the growing-map chain therefore creeps (2 mm, 0.02 degrees per frame); the scans are still taken at every pose
        dR = [synth.rot_from_rpy(0.0, 0.0, np.deg2rad(0.02) * (k % 2 * 2 - 1)) for k in range(K)]
        dt = [np.array([0.002, 0.001 * (k % 3 - 1), 0.0]) for k in range(K)]
    Rt, tt = [R0], [t0]
    for k in range(K):
        Rt.append(Rt[-1] @ dR[k]); tt.append(tt[-1] + dt[k])
    chunks = [synth.lidar_scan(rng, scene, R0, t0, extR, extT, min(map_rays, 400000), c["dept_err"], c["beam_err"], synth.AVIA["blind"], full_sphere) for _ in range(max(1, map_rays // 400000 + 1))]
    pw0, var0 = synth.world_points_and_var(np.concatenate(chunks)[:map_rays], R0, t0, extR, extT, P0, c["dept_err"], c["beam_err"])
    scans = []
    for k in range(K):
        s = synth.voxel_grid_downsample(synth.lidar_scan(rng, scene, Rt[k + 1], tt[k + 1], extR, extT, n_raw, c["dept_err"], c["beam_err"], synth.AVIA["blind"], full_sphere), downsample)
        if max_points and len(s) > max_points:
            s = s[np.sort(rng.permutation(len(s))[:max_points])]
        scans.append(np.ascontiguousarray(s, np.float32))
    motion = np.array([np.concatenate([(dR[k] @ synth.so3_exp(rng.normal(0, np.deg2rad(0.1), 3))).ravel(), dt[k] + rng.normal(0, 0.005, 3)]) for k in range(K)])
    q = np.concatenate([np.full(3, 1e-5), np.full(3, 1e-4), np.zeros(13)])
    # visual side: the map as it stands when frame k arrives, in the SAME room, around the TRUE pose of frame k (the scan points `pg` and new_frame_->T_f_w_ are
    # NOT inputs: the runner takes them from the LIO posterior)
    vs = synth.visual_scenario(seed=seed + 2, n_patches=4, L=L)              # camera, extrinsics, vio config
    cs, growth = [], None
    for k in range(1 if grow else K):
        ck = synth.retrieve_chain_scenario(seed=seed + 10 + k, n_pg=64, n_vis=n_vis, L=L, grid_n_height=34, normal_en=True, scene=scene, R0=Rt[k + 1], t0=tt[k + 1])
        ck.sel.active[:] = 1
        ck.sel.pg = None; ck.sel.R_cur = None; ck.sel.t_cur = None           # run-time quantities of the chain
        cs.append(ck)
    if grow:
        # scripts of the map maintenance that runs AFTER frame k's visual update (generateVisualMapPoints / updateVisualMapPoints / updateReferencePatch are the last
        # steps of processFrame): made at the true camera pose of frame k with frame k's image; frame k + 1 retrieves from the map they leave
        Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
        grng = np.random.default_rng(seed + 77)
        gm = GrowingMap(cs[0])
        imgs = [cs[0].img] + [np.clip(cs[0].img.astype(np.int32) + grng.integers(-2, 3, cs[0].img.shape), 0, 255).astype(np.uint8) for _ in range(K - 1)]
        scripts = []
        for k in range(K - 1):
            R_fw = Rci @ Rt[k + 1].T
            t_fw = -Rci @ Rt[k + 1].T @ tt[k + 1] + Pci
            sk = gm.generate(grng, R_fw, t_fw, imgs[k], frame_id=1000 + k, n_new=grow, n_touch=grow)
            gm.apply(sk)                                                      # (structure only matters here: the lists the next script reads)
            scripts.append(sk)
        growth = dict(scripts=scripts, imgs=imgs)
    return dict(c=c, extR=extR, extT=extT, R0=R0, t0=t0, P0=P0, pw0=pw0, var0=var0, scans=scans, motion=motion, q=q, cs=cs, vs=vs, L=L,
                R_true=Rt[1:], t_true=tt[1:], grow=growth)


def write_live_dir(d, live):
    os.makedirs(d, exist_ok=True)
    w = lambda name, arr: np.ascontiguousarray(arr).tofile(os.path.join(d, name + ".bin"))
    c, extR, extT, scans, cs, vs, L = live["c"], live["extR"], live["extT"], live["scans"], live["cs"], live["vs"], live["L"]
    pw0, var0, motion, q, R0, t0, P0 = live["pw0"], live["var0"], live["motion"], live["q"], live["R0"], live["t0"], live["P0"]
    K = len(scans)
    w("seq_bld_pw", pw0); w("seq_bld_var", var0.reshape(-1, 9))
    w("seq_map_cfg", np.array([c["voxel_size"], c["max_layer"], c["max_points_num"], c["min_eigen_value"]] + list(c["layer_init_num"])[:5], np.float64))
    w("seq_lidar_cfg", np.concatenate([[c["max_iterations"], c["max_layer"], c["sigma_num"], c["dept_err"], c["beam_err"], c["voxel_size"]], extR.ravel(), extT]).astype(np.float64))
    w("seq_scans", np.concatenate(scans).astype(np.float32)); w("seq_counts", np.array([len(s) for s in scans], np.int32))
    w("seq_motion", motion); w("seq_q", q)
    w("seq_state0", _state_vec(R0, t0, P0))
    w("vis_cfg", np.concatenate([[vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"], vs.cam["width"], vs.cam["height"], vs.cfg["img_point_cov"], L, vs.cfg["max_iterations"], 1.0],
                                 vs.Rcl.ravel(), vs.Pcl, vs.extR.ravel(), vs.extT]).astype(np.float64))
    c0 = cs[0]
    w("chain_cfg", np.array([K, c0.cfg["normal_en"], c0.cfg["ncc_en"], c0.cfg["ncc_thre"], c0.cfg["outlier_threshold"], L, c0.sel.border, c0.sel.grid_n_height], np.float64))
    f64, i32 = (lambda a: np.ascontiguousarray(a, np.float64)), (lambda a: np.ascontiguousarray(a, np.int32))
    for k, ck in enumerate(cs):
        pre = "chain%d_" % k
        w(pre + "img", ck.img); w(pre + "ref_imgs", ck.ref_imgs)
        for name, arr in (("pos", f64(ck.sel.pos)), ("normal", f64(ck.normal)), ("keys", np.ascontiguousarray(ck.sel.keys, np.int64)),
                          ("ninit", np.ascontiguousarray(ck.normal_initialized, np.uint8)), ("ref_patch", i32(ck.ref_patch)), ("obs_offset", i32(ck.obs_offset)),
                          ("obs_id", i32(ck.obs_id)), ("obs_img_idx", i32(ck.obs_img_idx)), ("obs_level", i32(ck.obs_level)), ("obs_px", f64(ck.obs_px)), ("obs_f", f64(ck.obs_f)),
                          ("obs_R", f64(ck.obs_R)), ("obs_t", f64(ck.obs_t)), ("obs_inv_expo", f64(ck.obs_inv_expo)), ("obs_patch", np.ascontiguousarray(ck.obs_patch, np.float32))):
            w(pre + name, arr)
    g = live.get("grow")
    if g:
        w("grow_cfg", np.array([len(g["scripts"])], np.int32))
        for k, im in enumerate(g["imgs"]):
            if k:
                w("grow%d_img" % k, im)
        for k, sk in enumerate(g["scripts"]):
            pre = "grow%d_" % k
            w(pre + "new_pos", f64(sk["new_pos"])); w(pre + "new_keys", np.ascontiguousarray(sk["new_keys"], np.int64)); w(pre + "new_normal", f64(sk["new_normal"]))
            w(pre + "ref_img", sk["img"])                                        # the image the new Features point at (= the image of the frame they were made in)
            for name in OBS_KEYS:
                a = sk["obs"][name]
                w(pre + "obs_" + name, np.ascontiguousarray(a, np.float32) if name == "patch" else (i32(a) if name in ("id", "img_idx", "level") else f64(a)))
            for name in ("t_point", "t_pop", "t_push", "t_ref", "t_flip", "t_toggle", "t_remove"):
                w(pre + name, i32(sk[name]))
    return dict(frames=K, points_per_scan=[len(s) for s in scans], visual_points=int(len(cs[0].sel.pos)), observations=int(len(cs[0].obs_id)))
