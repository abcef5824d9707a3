"""Load tests/golden/*.npz back into scenario-like objects."""
import os
from types import SimpleNamespace

import numpy as np

from scenarios.synth import FlatMap

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FM_FIELDS = ["root_key", "root_node", "root_center", "root_quarter", "node_plane", "node_child", "plane_normal", "plane_center", "plane_var", "plane_d", "plane_radius"]


def lidar_small():
    g = np.load(os.path.join(HERE, "lidar_small.npz"))
    fm = FlatMap(float(g["voxel_size"]), int(g["max_layer"]), *[g[f] for f in FM_FIELDS])
    cfg = {str(k): float(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    for k in ("max_iterations", "max_layer", "max_points_num"):
        if k in cfg:
            cfg[k] = int(cfg[k])
    sc = SimpleNamespace(fmap=fm, xyz=g["xyz"], R_prior=g["R_prior"], t_prior=g["t_prior"], P=g["P"], extR=g["extR"], extT=g["extT"], cfg=cfg)
    return sc, g


def visual_small():
    g = np.load(os.path.join(HERE, "visual_small.npz"))
    cam = dict(zip(("fx", "fy", "cx", "cy", "width", "height"), g["cam"]))
    cam["width"], cam["height"] = int(cam["width"]), int(cam["height"])
    cfg = dict(img_point_cov=float(g["img_point_cov"]), patch_pyrimid_level=int(g["L"]), max_iterations=int(g["max_iterations"]))
    vs = SimpleNamespace(img=g["img"], pos=g["pos"], warp_patch=g["warp_patch"], search_levels=g["search_levels"], inv_expo_list=g["inv_expo_list"],
                         R_prior=g["R_prior"], t_prior=g["t_prior"], tau_prior=float(g["tau_prior"]), P=g["P"], extR=g["extR"], extT=g["extT"], Rcl=g["Rcl"], Pcl=g["Pcl"],
                         cam=cam, cfg=cfg)
    return vs, g
