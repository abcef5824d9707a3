"""Inputs of fast-livo2_amd/host/live_chain (one chained live LIO + VIO frame through the C++ shim): a dump directory with
  seq_*   : first sweep (world points + covariances) for BuildVoxelMap, F down-sampled scans along a short path, the commanded motion between them, state0
  chain_* : a visual map (points, observations, reference images), the current image, the scan points of the visual scene, the frame pose
  vis_cfg : camera + extrinsics
in the binary layout fast-livo2_amd/host/shim_demo.cpp reads (tests/test_sequence_gpu.py, tests/test_host_shim_gpu.py write the same files for the parity tests)."""
import os

import numpy as np

from scenarios import synth


def _state_vec(R, t, P, inv_expo=1.0):
    return np.concatenate([np.asarray(R, float).ravel(), np.asarray(t, float), [inv_expo], np.zeros(12), np.asarray(P, float).ravel()])


SIZES = {
    "avia": dict(),                                                           # 24 000 rays per scan (~12.5 k points after the 0.1 m filter), 30 000 visual points
    "c4": dict(n_raw=620000, max_points=200000, room=(60.0, 60.0, 10.0), n_boxes=24, full_sphere=True, map_rays=1600000, n_vis=120000, n_pg=100000, n_frames=6),
}


def make_live(n_frames=12, n_raw=24000, map_rays=300000, room=(20.0, 20.0, 6.0), n_boxes=8, full_sphere=False, downsample=0.1, max_points=None,
              n_pg=10000, n_vis=30000, seed=191):
    """the sequence as Python objects (deterministic in its arguments): used by write_live_dir and by bench.py's CPU leg, which runs the oracle over the same frames"""
    rng = np.random.default_rng(seed)
    c = dict(synth.AVIA["lio"])
    extR, extT = synth.AVIA["extrinsic_R"], synth.AVIA["extrinsic_T"]
    scene = synth.make_room(rng, room, n_boxes)
    R0, t0 = scene.R_ws @ synth.rot_from_rpy(0.01, -0.015, 0.4), scene.R_ws @ np.array([0.3, -0.2, 1.4]) + scene.t_ws
    P0 = synth.default_cov() * 1e-3
    K = n_frames
    dR = [synth.rot_from_rpy(0.0, 0.0, 0.02 * (k % 2 * 2 - 1)) for k in range(K)]
    dt = [np.array([0.06, 0.02 * (k % 3 - 1), 0.0]) for k in range(K)]
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
        scans.append(s)
    motion = np.array([np.concatenate([(dR[k] @ synth.so3_exp(rng.normal(0, np.deg2rad(0.1), 3))).ravel(), dt[k] + rng.normal(0, 0.005, 3)]) for k in range(K)])
    q = np.concatenate([np.full(3, 1e-5), np.full(3, 1e-4), np.zeros(13)])
    # visual side
    L = 2
    cs = synth.retrieve_chain_scenario(seed=seed + 1, n_pg=n_pg, n_vis=n_vis, L=L, grid_n_height=34, normal_en=True)
    cs.sel.active[:] = 1
    vs = synth.visual_scenario(seed=seed + 2, n_patches=4, L=L)
    Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    R_cur, t_cur = cs.sel.R_cur, cs.sel.t_cur
    vs_c = synth.visual_scenario(seed=seed + 2, n_patches=4, L=L, R_true=R_cur.T @ Rci, t_true=R_cur.T @ (Pci - t_cur))
    return dict(c=c, extR=extR, extT=extT, R0=R0, t0=t0, P0=P0, pw0=pw0, var0=var0, scans=scans, motion=motion, q=q, cs=cs, vs=vs, vs_c=vs_c, L=L)


def write_live_dir(d, live):
    os.makedirs(d, exist_ok=True)
    w = lambda name, arr: np.ascontiguousarray(arr).tofile(os.path.join(d, name + ".bin"))
    c, extR, extT, scans, cs, vs, vs_c, L = live["c"], live["extR"], live["extT"], live["scans"], live["cs"], live["vs"], live["vs_c"], live["L"]
    pw0, var0, motion, q, R0, t0, P0 = live["pw0"], live["var0"], live["motion"], live["q"], live["R0"], live["t0"], live["P0"]
    R_cur, t_cur = cs.sel.R_cur, cs.sel.t_cur
    K = len(scans)
    w("seq_bld_pw", pw0); w("seq_bld_var", var0.reshape(-1, 9))
    w("seq_map_cfg", np.array([c["voxel_size"], c["max_layer"], c["max_points_num"], c["min_eigen_value"]] + list(c["layer_init_num"])[:5], np.float64))
    w("seq_lidar_cfg", np.concatenate([[c["max_iterations"], c["max_layer"], c["sigma_num"], c["dept_err"], c["beam_err"], c["voxel_size"]], extR.ravel(), extT]).astype(np.float64))
    w("seq_scans", np.concatenate(scans).astype(np.float32)); w("seq_counts", np.array([len(s) for s in scans], np.int32))
    w("seq_motion", motion); w("seq_q", q)
    w("seq_state0", _state_vec(R0, t0, P0))
    w("vis_cfg", np.concatenate([[vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"], vs.cam["width"], vs.cam["height"], vs.cfg["img_point_cov"], L, vs.cfg["max_iterations"], 1.0],
                                 vs.Rcl.ravel(), vs.Pcl, vs.extR.ravel(), vs.extT]).astype(np.float64))
    sv = _state_vec(vs_c.R_prior, vs_c.t_prior, vs_c.P, getattr(vs_c, "tau_prior", 1.0))
    w("chain_state_in", sv); w("chain_state_prop", sv)
    w("chain_cfg", np.concatenate([R_cur.ravel(), t_cur, [cs.inv_expo_cur, cs.cfg["normal_en"], cs.cfg["ncc_en"], cs.cfg["ncc_thre"], cs.cfg["outlier_threshold"], L, cs.sel.border,
                                                         cs.sel.grid_n_height]]).astype(np.float64))
    w("chain_img", cs.img); w("chain_ref_imgs", cs.ref_imgs)
    f64, i32 = (lambda a: np.ascontiguousarray(a, np.float64)), (lambda a: np.ascontiguousarray(a, np.int32))
    for name, arr in (("pg", f64(cs.sel.pg)), ("pos", f64(cs.sel.pos)), ("normal", f64(cs.normal)), ("keys", np.ascontiguousarray(cs.sel.keys, np.int64)),
                      ("ninit", np.ascontiguousarray(cs.normal_initialized, np.uint8)), ("ref_patch", i32(cs.ref_patch)), ("obs_offset", i32(cs.obs_offset)),
                      ("obs_id", i32(cs.obs_id)), ("obs_img_idx", i32(cs.obs_img_idx)), ("obs_level", i32(cs.obs_level)), ("obs_px", f64(cs.obs_px)), ("obs_f", f64(cs.obs_f)),
                      ("obs_R", f64(cs.obs_R)), ("obs_t", f64(cs.obs_t)), ("obs_inv_expo", f64(cs.obs_inv_expo)), ("obs_patch", np.ascontiguousarray(cs.obs_patch, np.float32))):
        w("chain_" + name, arr)
    return dict(frames=K, points_per_scan=[len(s) for s in scans], visual_points=int(len(cs.sel.pos)), observations=int(len(cs.obs_id)), scan_points_visual=int(len(cs.sel.pg)))
