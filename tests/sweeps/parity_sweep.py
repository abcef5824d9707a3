#!/usr/bin/env python
"""Parity sweep on a GPU box: full LiDAR and visual ESIKF updates through the C ABI against the oracle over many seeded scenarios.
Prints one line per scenario and a summary (matched-plane flips, float32 residual mismatches, worst accumulated delta-x relative error,
worst covariance relative error).  Usage: python tests/sweeps/parity_sweep.py [n_lidar_seeds] [n_visual_seeds] > profiles/rNN_parity_sweep.txt"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402  (checker only)
from scenarios import synth  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    nl = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    nv = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    livo2 = importlib.import_module("fast-livo2_amd")
    ctx = livo2.Context(0)
    tot = dict(dec=0, flips=0, dis=0, dx=0.0, P=0.0, iters_differ=0)
    print("# LiDAR: seed points matched iters flips residual_mismatch dx_rel P_rel")
    for k in range(nl):
        seed = 300 + k
        ext = None if k % 3 else synth.rot_from_rpy(0.05 * k, -0.03 * k, 0.02 * k)
        sc = synth.lidar_scenario(seed=seed, n_points=20000, downsample=0.1, n_boxes=4 + k % 6, rot_sigma_deg=0.2 + 0.1 * (k % 5), pos_sigma=0.01 + 0.01 * (k % 4), extR=ext)
        om = orc.OracleMap.from_flat(sc.fmap)
        ocur, oprop = H.states(sc, orc.StatePOD)
        pcur, pprop = H.states(sc, livo2.State)
        ref = orc.lidar_state_estimation(om, orc.lidar_cfg(sc.cfg, sc.extR, sc.extT), sc.xyz, ocur, oprop)
        pcfg = H.lidar_cfg_product(sc)
        ctx.upload_map(sc.fmap)
        ctx.set_scan(sc.xyz, pcfg)
        res, pts = ctx.lidar_update(pcur, pprop, pcfg, want=("match_plane", "dis_to_plane"))
        flips = int((pts["match_plane"] != ref["match_plane"]).sum())
        dis = int((pts["dis_to_plane"] != ref["dis"]).sum())
        so, sp = orc.state_arrays(ref["state"]), orc.state_arrays(res.state)
        dx_ref = np.concatenate([so["t"] - sc.t_prior, (sc.R_prior.T @ so["R"] - np.eye(3)).ravel()])
        dx_gpu = np.concatenate([sp["t"] - sc.t_prior, (sc.R_prior.T @ sp["R"] - np.eye(3)).ravel()])
        dx, dP = H.relerr(dx_gpu, dx_ref), H.relerr(sp["P"], so["P"])
        tot["dec"] += len(sc.xyz) * res.n_iters
        tot["flips"] += flips
        tot["dis"] += dis
        tot["dx"] = max(tot["dx"], dx)
        tot["P"] = max(tot["P"], dP)
        tot["iters_differ"] += int(res.n_iters != ref["n_iters"])
        print(seed, len(sc.xyz), int((ref["match_plane"] >= 0).sum()), res.n_iters, flips, dis, f"{dx:.2e}", f"{dP:.2e}", flush=True)
    print(f"# LiDAR summary: {tot['dec']} point-iterations, {tot['flips']} matched-plane flips, {tot['dis']} float32 residual mismatches, "
          f"{tot['iters_differ']} iteration-count differences, worst dx rel {tot['dx']:.2e}, worst P rel {tot['P']:.2e}")
    print("# visual: seed patches steps steps_equal dR dt dtau dP_rel")
    worst = dict(R=0.0, t=0.0, P=0.0)
    steps_bad, rows = 0, 0
    for k in range(nv):
        seed = 400 + k
        vs = synth.visual_scenario(seed=seed, n_patches=2000, rot_sigma_deg=0.03 + 0.01 * (k % 4))
        ocur, oprop = H.states(vs, orc.StatePOD)
        pcur, pprop = H.states(vs, livo2.State)
        exposure = bool(k % 2 == 0)
        ref = orc.visual_update(orc.visual_cfg(vs, exposure=exposure), vs, ocur, oprop)
        ctx.set_frame(vs.img, vs.pos, vs.warp_patch, vs.search_levels, vs.inv_expo_list)
        res, _ = ctx.visual_update(pcur, pprop, H.visual_cfg_product(vs, exposure=exposure))
        a = [(t.level, t.iteration, t.accepted, t.n_meas) for t in ref["trace"]]
        b = [(res.steps[i].level, res.steps[i].iteration, res.steps[i].accepted, res.steps[i].n_meas) for i in range(res.n_steps)]
        d = H.state_diff(res.state, ref["state"])
        steps_bad += int(a != b)
        rows += sum(s[3] for s in a)
        for key in ("R", "t", "P"):
            worst[key] = max(worst[key], d[key])
        print(seed, len(vs.pos), len(a), int(a == b), f"{d['R']:.2e}", f"{d['t']:.2e}", f"{d['inv_expo']:.2e}", f"{d['P']:.2e}", flush=True)
    print(f"# visual summary: {rows} scalar residual rows, {steps_bad} scenarios with a different accept/revert sequence, worst dR {worst['R']:.2e}, "
          f"dt {worst['t']:.2e}, dP rel {worst['P']:.2e}")
    widened_rows(ctx, livo2)
    ctx.close()


def widened_rows(ctx, livo2):
    """plane fit, retrieval (selection, tail, whole chain), pre-stage, IMU propagation — several seeds each."""
    from tests import imu_inputs as IMU
    from tests import plane_groups as PG
    print("# plane fit (livo2_plane_fit_batch vs init_plane): seed groups planes decision_flips worst_normal_err worst_plane_var_rel")
    for seed in range(500, 505):
        pw, var, off, kinds = PG.make_groups(seed=seed, n_groups=600, big=(400, 3000))
        out = ctx.plane_fit_batch(pw, var, off, 0.0025)
        ref, _ = orc.init_plane_batch(pw, var, off, 0.0025)
        flips, wn, wv, planes = 0, 0.0, 0.0, 0
        for g in range(len(off) - 1):
            if ref[g].is_plane and abs(ref[g].min_eigen_value - 0.0025) < 1e-7:
                continue
            flips += int(out[g].is_plane != ref[g].is_plane)
            if ref[g].is_plane and out[g].is_plane:
                planes += 1
                gap = max(ref[g].mid_eigen_value - ref[g].min_eigen_value, 1e-12)
                wn = max(wn, float(np.linalg.norm(np.array(out[g].normal) - np.array(ref[g].normal))) / max(1.0, ref[g].max_eigen_value / gap))
                q = np.array(ref[g].plane_var)
                wv = max(wv, float(np.linalg.norm(np.array(out[g].plane_var) - q) / np.linalg.norm(q)) / max(1.0, ref[g].max_eigen_value / gap))
        print(seed, len(off) - 1, planes, flips, f"{wn:.2e}", f"{wv:.2e}", flush=True)
    print("# retrieval tail (livo2_visual_retrieve_warp vs orc_warp): seed candidates accepted accept_mismatch search_level_mismatch patch_mismatch error_mismatch")
    for k, seed in enumerate(range(510, 515)):
        rs = synth.retrieve_scenario(seed=seed, n_cand=2000, normal_en=bool(k % 2 == 0), ncc_en=bool(k % 3 == 0), ncc_thre=0.9)
        ref, out = orc.warp_candidates(rs), ctx.retrieve_warp(rs)
        print(seed, len(rs.pos), int(ref["accepted"].sum()), int((out["accepted"] != ref["accepted"]).sum()), int((out["search_level"] != ref["search_level"]).sum()),
              int((out["patch_wrap"] != ref["patch_wrap"]).sum()), int((out["error"] != ref["error"]).sum()), flush=True)
    print("# retrieval selection (livo2_visual_select vs orc_select): seed scan_points visual_points cells cell_mismatch dist_mismatch discont_mismatch in_fov_mismatch")
    for seed in range(520, 525):
        ss = synth.select_scenario(seed=seed, n_pg=10000, n_vis=30000)
        ref = orc.visual_select(ss)
        ctx.visual_map_upload(ss.pos, ss.keys, ss.active)
        out = ctx.visual_select(ss)
        print(seed, len(ss.pg), len(ss.pos), int((ref["cell_point"] >= 0).sum()), int((out["cell_point"] != ref["cell_point"]).sum()), int((out["cell_dist"] != ref["cell_dist"]).sum()),
              int((out["discont"].astype(np.int32) != ref["discont"]).sum()), int((out["in_fov"].astype(np.int32) != ref["in_fov"]).sum()), flush=True)
    print("# retrieval chain (livo2_visual_retrieve_from_map vs orc select -> choice -> tail): seed normal_en cells candidates accepted cell_obs_mismatch ref_patch_mismatch "
          "candidate_order_mismatch patch_mismatch error_mismatch submap_mismatch")
    for k, seed in enumerate(range(525, 530)):
        cs = synth.retrieve_chain_scenario(seed=seed, n_pg=10000, n_vis=20000, grid_n_height=(17, 51, 102)[k % 3], normal_en=bool(k % 2 == 0), ncc_en=bool(k % 3 == 0), ncc_thre=0.8)
        ref = orc.visual_retrieve(cs)
        ctx.visual_map_upload(cs.sel.pos, cs.sel.keys, cs.sel.active)
        ctx.visual_obs_upload(cs)
        out = ctx.visual_retrieve_from_map(cs)
        same_n = out["n_candidates"] == len(ref["cand_cell"])
        sub_bad = int(out["n_accepted"] != len(ref["sub_point"]) or not np.array_equal(out["sub_point"], ref["sub_point"]) or not np.array_equal(out["sub_obs"], ref["sub_obs"]))
        print(seed, int(cs.cfg["normal_en"]), len(out["cell_point"]), out["n_candidates"], out["n_accepted"], int((out["cell_obs"] != ref["cell_obs"]).sum()),
              int((out["ref_patch"] != ref["ref_patch"]).sum()), int(not same_n or not np.array_equal(out["cand_cell"], ref["cand_cell"])),
              int((out["tail"]["patch_wrap"] != ref["tail"]["patch_wrap"]).sum()) if same_n else -1, int((out["tail"]["error"] != ref["tail"]["error"]).sum()) if same_n else -1,
              sub_bad, flush=True)
    print("# pre-stage (livo2_lidar_preprocess_scan vs orc_preprocess): seed raw_points undistorted_coords_differing max_ulp feats_down voxel_grid_mismatch")
    for k, seed in enumerate(range(530, 535)):
        rs = synth.raw_scan_scenario(seed=seed, n_raw=24000, extR=None if k % 2 else synth.rot_from_rpy(0.1, -0.05, 0.02 * k))
        c = livo2.LidarCfg()
        c.max_iterations, c.max_layer = 5, 2
        c.sigma_num, c.dept_err, c.beam_err, c.voxel_size, c.deg2rad = 3.0, 0.02, 0.05, 0.5, 0.017453293
        c.extR[:] = rs.extR.ravel().tolist(); c.extT[:] = rs.extT.tolist()
        nd, und, down = ctx.preprocess_scan(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.leaf, c)
        ref_u = orc.undistort(rs.xyz, rs.curvature, rs.poses, rs.rot_end, rs.pos_end, rs.extR, rs.extT)
        ulp = np.abs(und - ref_u) / np.spacing(np.maximum(np.abs(ref_u), 1e-3).astype(np.float32))
        ref_d = orc.voxel_grid(und, rs.leaf)
        print(seed, len(rs.xyz), int((und != ref_u).sum()), f"{ulp.max():.1f}", nd, int(nd != len(ref_d)) + (int((down != ref_d).sum()) if nd == len(ref_d) else 0), flush=True)
    print("# IMU propagation (livo2_imu_propagate vs orc_imu): seed samples dR dpos dvel dP_rel poses_abs")
    for seed, n in ((540, 20), (541, 40), (542, 200), (543, 7), (544, 1000)):
        steps = IMU.make_steps(seed, n=n)
        ref, rposes, _ = orc.imu_propagate(IMU.make_state(orc, orc.StatePOD, seed), steps, IMU.CFG)
        out, poses = ctx.imu_propagate(IMU.make_state(orc, livo2.State, seed), steps, orc.imu_cfg(IMU.CFG, cls=livo2.ImuCfg))
        a, b = orc.state_arrays(out), orc.state_arrays(ref)
        print(seed, n, f"{np.abs(a['R'] - b['R']).max():.2e}", f"{np.abs(a['t'] - b['t']).max():.2e}", f"{np.abs(a['vel'] - b['vel']).max():.2e}",
              f"{np.abs(a['P'] - b['P']).max() / np.abs(b['P']).max():.2e}", f"{np.abs(poses - rposes).max():.2e}", flush=True)


if __name__ == "__main__":
    main()
