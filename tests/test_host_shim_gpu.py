"""The C++ host shim (fast-livo2_amd/host): reference-shaped containers -> C ABI -> HIP, driven like LIVMapper drives the reference
managers, compared with the oracle."""
import os
import subprocess

import numpy as np
import pytest

from scenarios import synth
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "fast-livo2_amd", "lib", "shim_demo")


def _state_vec(s):
    return np.frombuffer(bytes(s), dtype=np.float64).copy()


def test_shim_builds_and_links():
    subprocess.run(["make", "-C", os.path.join(ROOT, "fast-livo2_amd", "host")], check=True, capture_output=True)
    assert os.path.exists(DEMO)
    out = subprocess.run(["ldd", DEMO], capture_output=True, text=True).stdout
    assert "liblivo2_host.so" in out and "liblivo2_hip.so" in out and "not found" not in out


@pytest.mark.gpu
def test_shim_matches_oracle(tmp_path, orc, livo2):
    d = str(tmp_path)
    sc = synth.lidar_scenario(seed=17, n_points=2500, downsample=0.1)
    fm = sc.fmap
    for name in ("root_key", "root_node", "root_center", "root_quarter", "node_plane", "node_child", "plane_normal", "plane_center", "plane_var", "plane_d", "plane_radius"):
        np.ascontiguousarray(getattr(fm, name)).tofile(os.path.join(d, name + ".bin"))
    sc.xyz.tofile(os.path.join(d, "xyz.bin"))
    c = sc.cfg
    np.concatenate([[c["max_iterations"], c["max_layer"], c["sigma_num"], c["dept_err"], c["beam_err"], c["voxel_size"]], sc.extR.ravel(), sc.extT]).astype(np.float64).tofile(os.path.join(d, "lidar_cfg.bin"))
    ocur, oprop = H.states(sc, orc.StatePOD)
    _state_vec(ocur).tofile(os.path.join(d, "state_in.bin")); _state_vec(oprop).tofile(os.path.join(d, "state_prop.bin"))
    vs = synth.visual_scenario(seed=18, n_patches=100)
    vs.img.tofile(os.path.join(d, "img.bin"))
    np.concatenate([[vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"], vs.cam["width"], vs.cam["height"], vs.cfg["img_point_cov"], vs.cfg["patch_pyrimid_level"],
                     vs.cfg["max_iterations"], 1.0], vs.Rcl.ravel(), vs.Pcl, vs.extR.ravel(), vs.extT]).astype(np.float64).tofile(os.path.join(d, "vis_cfg.bin"))
    vs.pos.astype(np.float64).tofile(os.path.join(d, "vis_pos.bin")); vs.warp_patch.astype(np.float32).tofile(os.path.join(d, "vis_warp.bin"))
    vs.search_levels.astype(np.int32).tofile(os.path.join(d, "vis_search.bin")); vs.inv_expo_list.astype(np.float64).tofile(os.path.join(d, "vis_invexpo.bin"))
    vcur, vprop = H.states(vs, orc.StatePOD)
    _state_vec(vcur).tofile(os.path.join(d, "vis_state_in.bin")); _state_vec(vprop).tofile(os.path.join(d, "vis_state_prop.bin"))

    # FitPlanes leg: re-fit 60 planes of the map from fresh point groups near them, then run the update again on the refreshed map
    from tests import plane_groups as PG
    rng = np.random.default_rng(5)
    fit_idx = rng.permutation(fm.n_planes)[:60].astype(np.int32)
    fpts, fvar, foff = [], [], [0]
    for p in fit_idx:
        n = int(rng.integers(8, 40))
        Q, _ = np.linalg.qr(np.c_[fm.plane_normal[p], rng.normal(size=(3, 2))])
        q = (rng.normal(size=(n, 3)) * np.array([0.004, 0.12, 0.1])) @ Q.T + fm.plane_center[p]
        fpts.append(q.astype(np.float32).astype(np.float64)); fvar.append(PG.random_spd(rng, n)); foff.append(foff[-1] + n)
    fpts, fvar, foff = np.concatenate(fpts), np.concatenate(fvar).reshape(-1, 9), np.array(foff, np.int32)
    fit_idx.tofile(os.path.join(d, "fit_plane.bin")); fpts.tofile(os.path.join(d, "fit_pw.bin")); fvar.tofile(os.path.join(d, "fit_var.bin")); foff.tofile(os.path.join(d, "fit_off.bin"))
    # retrieval leg
    # (L = 2 and a 72-px margin keep every patch of the later visual update inside the image: the oracle, like the reference, reads out of bounds otherwise)
    rs = synth.retrieve_scenario(seed=19, n_cand=300, L=2, margin=72)
    np.concatenate([rs.R_cur.ravel(), rs.t_cur, [rs.inv_expo_cur, rs.cfg["normal_en"], rs.cfg["ncc_en"], rs.cfg["ncc_thre"], rs.cfg["outlier_threshold"], 2]]).astype(np.float64).tofile(os.path.join(d, "retr_cfg.bin"))
    # the IMU state whose camera pose is the retrieval's new_frame_->T_f_w_ (Rcw = Rci Rwi^T, Pcw = -Rcw Pwi + Pci, vio.cpp:1540-1543)
    Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    vs_r = synth.visual_scenario(seed=18, n_patches=4, L=2, R_true=rs.R_cur.T @ Rci, t_true=rs.R_cur.T @ (Pci - rs.t_cur))
    rcur, rprop = H.states(vs_r, orc.StatePOD)
    _state_vec(rcur).tofile(os.path.join(d, "retr_state_in.bin")); _state_vec(rprop).tofile(os.path.join(d, "retr_state_prop.bin"))
    rs.img.tofile(os.path.join(d, "retr_img.bin")); rs.ref_imgs.tofile(os.path.join(d, "retr_ref_imgs.bin"))
    for name in ("pos", "normal", "ref_px", "ref_f", "ref_R", "ref_t", "ref_inv_expo"):
        np.ascontiguousarray(getattr(rs, name), np.float64).tofile(os.path.join(d, "retr_" + name + ".bin"))
    rs.ref_img_idx.astype(np.int32).tofile(os.path.join(d, "retr_ref_img_idx.bin")); rs.ref_level.astype(np.int32).tofile(os.path.join(d, "retr_ref_level.bin"))

    # ImuProcess::ForwardPropagate, then VoxelMapManager::UndistortAndDownsample with the poses it produced
    from tests import imu_inputs as IMU
    isteps = IMU.make_steps(7, n=20)
    ist = IMU.make_state(orc, orc.StatePOD, 7)
    isteps.tofile(os.path.join(d, "imu_steps.bin")); _state_vec(ist).tofile(os.path.join(d, "imu_state_in.bin"))
    mean_acc = np.array([0.0, 0.0, -IMU.CFG["mean_acc_norm"]])
    np.concatenate([IMU.CFG["cov_gyr"], IMU.CFG["cov_acc"], IMU.CFG["cov_bias_gyr"], IMU.CFG["cov_bias_acc"], [IMU.CFG["cov_inv_expo"]], mean_acc, [1, 1, 1]]).astype(np.float64).tofile(os.path.join(d, "imu_cfg.bin"))
    raw = synth.raw_scan_scenario(seed=20, n_raw=8000)
    raw.xyz.tofile(os.path.join(d, "raw_xyz.bin")); raw.curvature.tofile(os.path.join(d, "raw_curvature.bin"))
    np.concatenate([raw.extR.ravel(), raw.extT, [raw.leaf, 0.5]]).astype(np.float64).tofile(os.path.join(d, "raw_cfg.bin"))
    # selection half of the retrieval
    ss = synth.select_scenario(seed=21, n_pg=5000, n_vis=4000)
    ss.pos.tofile(os.path.join(d, "sel_pos.bin")); ss.keys.astype(np.int64).tofile(os.path.join(d, "sel_keys.bin")); ss.active.astype(np.uint8).tofile(os.path.join(d, "sel_active.bin"))
    ss.pg.tofile(os.path.join(d, "sel_pg.bin"))
    np.concatenate([ss.R_cur.ravel(), ss.t_cur, [ss.border, ss.grid_n_height]]).astype(np.float64).tofile(os.path.join(d, "sel_cfg.bin"))

    r = subprocess.run([DEMO, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr

    om = orc.OracleMap.from_flat(fm)
    ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, ocur, oprop)
    out = np.fromfile(os.path.join(d, "out_state.bin"))
    refv = _state_vec(ref["state"])
    assert np.allclose(out[:25], refv[:25], rtol=0, atol=1e-9) and H.relerr(out[25:], refv[25:]) < 1e-8
    assert int(np.fromfile(os.path.join(d, "out_effct.bin"), dtype=np.int32)[0]) == int((ref["match_plane"] >= 0).sum())
    assert np.array_equal(np.fromfile(os.path.join(d, "out_ptpl_dis.bin"), dtype=np.float32), ref["dis"][ref["match_plane"] >= 0])      # ptpl_list_ keeps point order
    assert np.array_equal(np.fromfile(os.path.join(d, "out_pv_normal.bin")).reshape(-1, 3), ref["normal"])
    assert H.relerr(np.fromfile(os.path.join(d, "out_pv_var.bin")).reshape(-1, 9), ref["var"]) < 1e-13
    # geoQuat_ = createQuaternionMsgFromRollPitchYaw(RotMtoEuler(state_.rot_end)) (voxel_map.cpp:493): the quaternion of the posterior rotation, up to sign
    from scipy.spatial.transform import Rotation
    q, qr = np.fromfile(os.path.join(d, "out_geoquat.bin")), Rotation.from_matrix(out[:9].reshape(3, 3)).as_quat()
    assert min(np.abs(q - qr).max(), np.abs(q + qr).max()) < 1e-12

    vref = orc.visual_update(orc.visual_cfg(vs, num_threads=4), vs, vcur, vprop)           # the shim's default mp_proc_num = 4 (MP_PROC_NUM of the reference build)
    tms = np.fromfile(os.path.join(d, "vis_out_times.bin"))
    assert 0 < tms[0] < 1e-2 and 0 < tms[1] < 1e-2                                          # compute_jacobian_time / update_ekf_time in seconds (vio.h:114)
    vout = np.fromfile(os.path.join(d, "vis_out_state.bin"))
    vrefv = _state_vec(vref["state"])
    assert np.allclose(vout[:25], vrefv[:25], rtol=0, atol=1e-9) and H.relerr(vout[25:], vrefv[25:]) < 1e-8
    assert np.array_equal(np.fromfile(os.path.join(d, "vis_out_errors.bin"), dtype=np.float32), vref["errors"])
    assert H.relerr(np.fromfile(os.path.join(d, "vis_out_G.bin")).reshape(19, 19), vref["G"]) < 1e-7

    # FitPlanes: fitted members vs the oracle's init_plane, and the second update vs the same refresh done through the Python ABI wrappers
    fits = [orc.init_plane(fpts[foff[g]:foff[g + 1]], fvar[foff[g]:foff[g + 1]], 0.0025) for g in range(len(fit_idx))]
    assert np.array_equal(np.fromfile(os.path.join(d, "fit_out_is_plane.bin"), dtype=np.int32), np.array([f.is_plane for f in fits]))
    assert np.allclose(np.fromfile(os.path.join(d, "fit_out_center.bin")).reshape(-1, 3), np.array([list(f.center) for f in fits]), rtol=1e-13)
    assert np.allclose(np.fromfile(os.path.join(d, "fit_out_normal.bin")).reshape(-1, 3), np.array([list(f.normal) for f in fits]), atol=1e-6)
    assert np.allclose(np.fromfile(os.path.join(d, "fit_out_radius.bin"), dtype=np.float32), np.array([f.radius for f in fits]), rtol=1e-5)
    ctx = livo2.Context(0)
    ctx.upload_map(fm)
    ctx.update_planes(fit_idx, np.array([list(f.normal) for f in fits]), np.array([list(f.center) for f in fits]), np.array([list(f.plane_var) for f in fits]),
                      np.array([f.d for f in fits], np.float32), np.array([f.radius for f in fits], np.float32))
    pcur, pprop = H.states(sc, livo2.State)
    ctx.set_scan(sc.xyz, H.lidar_cfg_product(sc))
    res2, _ = ctx.lidar_update(pcur, pprop, H.lidar_cfg_product(sc))
    out2 = np.fromfile(os.path.join(d, "out_state2.bin"))
    ref2 = _state_vec(res2.state)
    assert np.allclose(out2[:25], ref2[:25], rtol=0, atol=1e-6) and H.relerr(out2[25:], ref2[25:]) < 1e-4
    assert np.abs(out2[:25] - out[:25]).max() > 1e-9          # the refreshed planes did change the estimate
    ctx.close()

    # retrieval: survivors (order, errors, search levels) vs the oracle, visual update on the resident frame vs the oracle on the survivors
    wref = orc.warp_candidates(rs)
    keep = np.nonzero(wref["accepted"])[0]
    assert np.array_equal(np.fromfile(os.path.join(d, "retr_out_kept.bin"), dtype=np.int32), keep)
    assert np.array_equal(np.fromfile(os.path.join(d, "retr_out_errors.bin"), dtype=np.float32), wref["error"][keep])
    assert np.array_equal(np.fromfile(os.path.join(d, "retr_out_search.bin"), dtype=np.int32), wref["search_level"][keep])
    vs_r.img, vs_r.pos, vs_r.warp_patch, vs_r.search_levels, vs_r.inv_expo_list = rs.img, rs.pos[keep], wref["patch_wrap"][keep], wref["search_level"][keep], rs.ref_inv_expo[keep]
    rcur.inv_expo = rs.inv_expo_cur
    vref2 = orc.visual_update(orc.visual_cfg(vs_r, num_threads=4), vs_r, rcur, rprop)
    vout2 = np.fromfile(os.path.join(d, "retr_out_state.bin"))
    vref2v = _state_vec(vref2["state"])
    assert np.allclose(vout2[:25], vref2v[:25], rtol=0, atol=1e-8) and H.relerr(vout2[25:], vref2v[25:]) < 1e-7

    # IMU propagation and the pre-stage through the shim
    iref, iposes, _ = orc.imu_propagate(ist, isteps, dict(IMU.CFG, first_call=1))          # the shim's first ForwardPropagate: imu_time_init is false, tau = 1.0
    iout = np.fromfile(os.path.join(d, "imu_out_state.bin"))
    irefv = _state_vec(iref)
    assert np.allclose(iout[:25], irefv[:25], rtol=0, atol=1e-12) and H.relerr(iout[25:], irefv[25:]) < 1e-12
    poses = np.fromfile(os.path.join(d, "imu_out_poses.bin")).reshape(-1, 22)
    assert len(poses) == len(isteps) + 1 and np.abs(poses[1:] - iposes).max() < 1e-11
    so = orc.state_arrays(iref)
    und = orc.undistort(raw.xyz, raw.curvature, poses, so["R"], so["t"], raw.extR, raw.extT)
    down_ref = orc.voxel_grid(und, raw.leaf)
    down = np.fromfile(os.path.join(d, "raw_out_down.bin"), dtype=np.float32).reshape(-1, 3)
    assert down.shape == down_ref.shape and np.abs(down - down_ref).max() < 1e-5

    # selection through the shim: the points the loop at vio.cpp:598 goes on with, in grid order
    sref = orc.visual_select(ss)
    keep_ref = [int(p) for p, dsc in zip(sref["cell_point"], sref["discont"]) if p >= 0 and not dsc]
    assert list(np.fromfile(os.path.join(d, "sel_out_kept.bin"), dtype=np.int32)) == keep_ref and len(keep_ref) > 50
    assert np.array_equal(np.fromfile(os.path.join(d, "sel_out_map_dist.bin"), dtype=np.float32), sref["cell_dist"])


@pytest.mark.gpu
@pytest.mark.parametrize("normal_en", [True, False])
def test_shim_retrieve_from_visual_sparse_map(tmp_path, orc, normal_en):
    """VIOManager::retrieveFromVisualSparseMap through the shim (feat_map of VisualPoints with their Feature lists -> device mirror -> one chain
    of launches), then computeJacobianAndUpdateEKF on the resident sub-map — against the chained oracle."""
    d = str(tmp_path)
    L = 2                                           # like the retrieval leg above: keeps every patch of the later update inside the image
    cs = synth.retrieve_chain_scenario(seed=91, n_pg=6000, n_vis=7000, L=L, grid_n_height=34, normal_en=normal_en)
    cs.sel.active[:] = 1                            # the shim derives `active` from obs_.size() > 0, and every point here has observations
    vs = synth.visual_scenario(seed=18, n_patches=4, L=L)
    Rci, Pci = synth.vio_constants(vs.extR, vs.extT, vs.Rcl, vs.Pcl)
    R_cur, t_cur = cs.sel.R_cur, cs.sel.t_cur
    vs_c = synth.visual_scenario(seed=18, n_patches=4, L=L, R_true=R_cur.T @ Rci, t_true=R_cur.T @ (Pci - t_cur))   # the IMU state whose camera pose is new_frame_->T_f_w_
    ccur, cprop = H.states(vs_c, orc.StatePOD)
    cs.img.tofile(os.path.join(d, "img.bin"))
    np.concatenate([[vs.cam["fx"], vs.cam["fy"], vs.cam["cx"], vs.cam["cy"], vs.cam["width"], vs.cam["height"], vs.cfg["img_point_cov"], L, vs.cfg["max_iterations"], 1.0],
                    vs.Rcl.ravel(), vs.Pcl, vs.extR.ravel(), vs.extT]).astype(np.float64).tofile(os.path.join(d, "vis_cfg.bin"))
    _state_vec(ccur).tofile(os.path.join(d, "vis_state_in.bin")); _state_vec(cprop).tofile(os.path.join(d, "vis_state_prop.bin"))
    _state_vec(ccur).tofile(os.path.join(d, "chain_state_in.bin")); _state_vec(cprop).tofile(os.path.join(d, "chain_state_prop.bin"))
    np.concatenate([R_cur.ravel(), t_cur, [cs.inv_expo_cur, cs.cfg["normal_en"], cs.cfg["ncc_en"], cs.cfg["ncc_thre"], cs.cfg["outlier_threshold"], L, cs.sel.border,
                                          cs.sel.grid_n_height]]).astype(np.float64).tofile(os.path.join(d, "chain_cfg.bin"))
    cs.img.tofile(os.path.join(d, "chain_img.bin")); cs.ref_imgs.tofile(os.path.join(d, "chain_ref_imgs.bin"))
    f64, i32 = (lambda a: np.ascontiguousarray(a, np.float64)), (lambda a: np.ascontiguousarray(a, np.int32))
    for name, arr in (("pg", f64(cs.sel.pg)), ("pos", f64(cs.sel.pos)), ("normal", f64(cs.normal)), ("keys", np.ascontiguousarray(cs.sel.keys, np.int64)),
                      ("ninit", np.ascontiguousarray(cs.normal_initialized, np.uint8)), ("ref_patch", i32(cs.ref_patch)), ("obs_offset", i32(cs.obs_offset)),
                      ("obs_id", i32(cs.obs_id)), ("obs_img_idx", i32(cs.obs_img_idx)), ("obs_level", i32(cs.obs_level)), ("obs_px", f64(cs.obs_px)), ("obs_f", f64(cs.obs_f)),
                      ("obs_R", f64(cs.obs_R)), ("obs_t", f64(cs.obs_t)), ("obs_inv_expo", f64(cs.obs_inv_expo)), ("obs_patch", np.ascontiguousarray(cs.obs_patch, np.float32))):
        arr.tofile(os.path.join(d, "chain_" + name + ".bin"))
    r = subprocess.run([DEMO, d], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "retrieveFromVisualSparseMap" in r.stdout

    ref = orc.visual_retrieve(cs)
    keep = ref["tail"]["accepted"] == 1
    assert keep.sum() > 40
    rd = lambda name, dt: np.fromfile(os.path.join(d, "chain_out_" + name + ".bin"), dtype=dt)
    assert np.array_equal(rd("kept", np.int32), ref["sub_point"])                       # visual_submap->voxel_points, in order
    assert np.array_equal(rd("errors", np.float32), ref["tail"]["error"][keep])
    assert np.array_equal(rd("search", np.int32), ref["tail"]["search_level"][keep])
    assert np.array_equal(rd("inv_expo", np.float64), cs.obs_inv_expo[ref["sub_obs"]])
    assert np.array_equal(rd("map_dist", np.float32), ref["sel"]["cell_dist"])
    assert np.array_equal(rd("ref_patch", np.int32), ref["ref_patch"])                  # pt->ref_patch / has_ref_patch_ written back
    if normal_en:
        assert np.array_equal(rd("kept_ref", np.int32), ref["sub_obs"])                 # with normal_en ref_ftr IS pt->ref_patch
    vs_c.img, vs_c.pos, vs_c.warp_patch = cs.img, cs.sel.pos[ref["sub_point"]], ref["tail"]["patch_wrap"][keep]
    vs_c.search_levels, vs_c.inv_expo_list = ref["tail"]["search_level"][keep], cs.obs_inv_expo[ref["sub_obs"]]
    ccur.inv_expo = cs.inv_expo_cur
    vref = orc.visual_update(orc.visual_cfg(vs_c, num_threads=4), vs_c, ccur, cprop)
    vout = rd("state", np.float64)
    vrefv = _state_vec(vref["state"])
    assert np.allclose(vout[:25], vrefv[:25], rtol=0, atol=1e-8) and H.relerr(vout[25:], vrefv[25:]) < 1e-7
