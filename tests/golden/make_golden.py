#!/usr/bin/env python
"""Generates tests/golden/*.npz: small frozen inputs + the oracle's outputs for them.

The reference (ROS1/PCL/Eigen/OpenCV/Sophus/vikit) cannot be built or imported here and ships no golden vectors for this path
(SURVEY.md §4, §8c), so these fixtures freeze the ORACLE's answers (golden build: -O2 -ffp-contract=off) on frozen inputs: they pin
the oracle against regressions and give the GPU tests reference values that do not depend on regenerating the scenario.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402
from scenarios import synth  # noqa: E402
from tests import helpers as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
FM_FIELDS = ["root_key", "root_node", "root_center", "root_quarter", "node_plane", "node_child", "plane_normal", "plane_center", "plane_var", "plane_d", "plane_radius"]


def lidar():
    sc = synth.lidar_scenario(seed=21, n_points=600, downsample=0.1, map_rays_factor=10)
    om = orc.OracleMap.from_flat(sc.fmap)
    cfg = orc.lidar_cfg(sc.cfg, sc.extR, sc.extT)
    cur, prop = H.states(sc, orc.StatePOD)
    it = orc.lidar_iterate(om, cfg, sc.xyz, cur, prop)
    full = orc.lidar_state_estimation(om, cfg, sc.xyz, cur, prop)
    so = orc.state_arrays(full["state"])
    np.savez_compressed(
        os.path.join(OUT, "lidar_small.npz"), xyz=sc.xyz, R_prior=sc.R_prior, t_prior=sc.t_prior, P=sc.P, extR=sc.extR, extT=sc.extT,
        cfg_keys=np.array(sorted(k for k in sc.cfg if k != "layer_init_num")), cfg_vals=np.array([float(sc.cfg[k]) for k in sorted(sc.cfg) if k != "layer_init_num"]),
        voxel_size=sc.fmap.voxel_size, max_layer=sc.fmap.max_layer, **{f: getattr(sc.fmap, f) for f in FM_FIELDS},
        it_HtH=it["HtH"], it_Htz=it["Htz"], it_n_eff=it["n_eff"], it_match=it["match_plane"], it_dis=it["dis"], it_pw=it["pw"], it_Rinv=it["Rinv"], it_Hrow=it["Hrow"],
        n_iters=full["n_iters"], tr_HtH=np.array([np.array(t.HtH) for t in full["trace"]]), tr_Htz=np.array([np.array(t.Htz) for t in full["trace"]]),
        tr_sol=np.array([np.array(t.solution) for t in full["trace"]]), tr_neff=np.array([t.n_eff for t in full["trace"]]),
        out_R=so["R"], out_t=so["t"], out_P=so["P"], out_match=full["match_plane"], out_dis=full["dis"])


def visual():
    vs = synth.visual_scenario(seed=22, n_patches=48)
    cfg = orc.visual_cfg(vs)
    cur, prop = H.states(vs, orc.StatePOD)
    it = orc.visual_iterate(cfg, vs, 1, cur)
    full = orc.visual_update(cfg, vs, cur, prop)
    so = orc.state_arrays(full["state"])
    np.savez_compressed(
        os.path.join(OUT, "visual_small.npz"), img=vs.img, pos=vs.pos, warp_patch=vs.warp_patch, search_levels=vs.search_levels, inv_expo_list=vs.inv_expo_list,
        R_prior=vs.R_prior, t_prior=vs.t_prior, tau_prior=vs.tau_prior, P=vs.P, extR=vs.extR, extT=vs.extT, Rcl=vs.Rcl, Pcl=vs.Pcl,
        cam=np.array([vs.cam[k] for k in ("fx", "fy", "cx", "cy", "width", "height")], float), img_point_cov=vs.cfg["img_point_cov"], L=vs.cfg["patch_pyrimid_level"],
        max_iterations=vs.cfg["max_iterations"], it_level=1, it_z=it["z"], it_H=it["H"], it_HtH=it["HtH"], it_Htz=it["Htz"], it_errors=it["errors"], it_error=it["error"],
        steps=np.array([(t.level, t.iteration, t.accepted, t.n_meas) for t in full["trace"]]), step_error=np.array([t.error for t in full["trace"]]),
        step_sol=np.array([np.array(t.solution) for t in full["trace"]]), out_R=so["R"], out_t=so["t"], out_P=so["P"], out_tau=so["inv_expo"], out_G=full["G"],
        out_errors=full["errors"])


def plane_fit():
    from tests import plane_groups as PG
    pw, var, off, kinds = PG.make_groups(seed=41, n_groups=24, big=(300,))
    out, _ = orc.init_plane_batch(pw, var, off, 0.0025)
    pick = lambda name, n: np.array([list(getattr(o, name)) for o in out]).reshape(len(out), n)
    np.savez_compressed(os.path.join(OUT, "plane_fit_small.npz"), pw=pw, var=var, off=off, thr=0.0025, center=pick("center", 3), normal=pick("normal", 3),
                        covariance=pick("covariance", 9), plane_var=pick("plane_var", 36), radius=np.array([o.radius for o in out], np.float32),
                        d=np.array([o.d for o in out], np.float32), eig=np.array([[o.min_eigen_value, o.mid_eigen_value, o.max_eigen_value] for o in out], np.float32),
                        is_plane=np.array([o.is_plane for o in out], np.int32))


def retrieve():
    rs = synth.retrieve_scenario(seed=42, n_cand=40, n_ref=2)
    # (the 640x512 images are not stored: the fixture keeps the scenario seed and an image checksum)
    o = orc.warp_candidates(rs)
    np.savez_compressed(os.path.join(OUT, "retrieve_small.npz"), seed=42, n_cand=40, n_ref=2, img_sum=int(rs.img.astype(np.int64).sum()),
                        accepted=o["accepted"], search_level=o["search_level"], error=o["error"], ncc=o["ncc"], A=o["A"],
                        patch_wrap_l0=o["patch_wrap"][:, 0], patch_wrap_sum=o["patch_wrap"].astype(np.float64).sum(axis=(1, 2)))


def retrieve_chain():
    """Whole retrieveFromVisualSparseMap (selection -> reference-patch choice -> tail) for both choice modes; like `retrieve`, the fixture keeps the
    scenario seed and checksums of the generated inputs instead of the images and the observation table."""
    keep = {}
    for mode, normal_en in (("n", True), ("c", False)):
        cs = synth.retrieve_chain_scenario(seed=43, n_pg=2500, n_vis=3000, grid_n_height=34, normal_en=normal_en)
        o = orc.visual_retrieve(cs)
        keep.update({mode + "_cell_point": o["sel"]["cell_point"], mode + "_cell_dist": o["sel"]["cell_dist"], mode + "_discont": o["sel"]["discont"],
                     mode + "_cell_obs": o["cell_obs"], mode + "_ref_patch": o["ref_patch"], mode + "_cand_cell": o["cand_cell"], mode + "_sub_point": o["sub_point"],
                     mode + "_sub_obs": o["sub_obs"], mode + "_search_level": o["tail"]["search_level"], mode + "_error": o["tail"]["error"],
                     mode + "_accepted": o["tail"]["accepted"], mode + "_A": o["tail"]["A"]})
    np.savez_compressed(os.path.join(OUT, "retrieve_chain_small.npz"), seed=43, n_pg=2500, n_vis=3000, grid_n_height=34, img_sum=int(cs.img.astype(np.int64).sum()),
                        patch_sum=float(cs.obs_patch.astype(np.float64).sum()), n_obs=int(cs.obs_offset[-1]), **keep)


if __name__ == "__main__":
    plane_fit()
    retrieve()
    retrieve_chain()
    lidar()
    visual()
    for f in ("lidar_small.npz", "visual_small.npz"):
        print(f, os.path.getsize(os.path.join(OUT, f)), "bytes")
