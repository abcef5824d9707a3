"""Shared helpers for the parity tests: build matching oracle / product inputs from one scenario."""
import importlib

import numpy as np

from oracle import orc


def product():
    return importlib.import_module("fast-livo2_amd")


def lidar_cfg_product(sc, max_iterations=None):
    livo2 = product()
    c = livo2.LidarCfg()
    c.max_iterations = int(max_iterations or sc.cfg["max_iterations"])
    c.max_layer = int(sc.cfg["max_layer"])
    c.sigma_num, c.dept_err, c.beam_err, c.voxel_size, c.deg2rad = float(sc.cfg["sigma_num"]), float(sc.cfg["dept_err"]), float(sc.cfg["beam_err"]), float(sc.cfg["voxel_size"]), 0.017453293
    c.extR[:] = sc.extR.ravel().tolist()
    c.extT[:] = sc.extT.tolist()
    return c


def states(sc, cls, R=None, t=None, inv_expo=1.0):
    """(state, prior) in the given struct class, both at the prior pose unless R/t are given for the iterate."""
    prior = orc.make_state(sc.R_prior, sc.t_prior, sc.P, inv_expo=getattr(sc, "tau_prior", 1.0), cls=cls)
    cur = orc.make_state(sc.R_prior if R is None else R, sc.t_prior if t is None else t, sc.P, inv_expo=getattr(sc, "tau_prior", inv_expo), cls=cls)
    return cur, prior


def visual_cfg_product(sc, exposure=True, max_iterations=None, inverse=False, mp_proc_num=1, distortion=None):
    livo2 = product()
    c = livo2.VisualCfg()
    c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = sc.cam["fx"], sc.cam["fy"], sc.cam["cx"], sc.cam["cy"]
    c.cam.distortion, c.cam.width, c.cam.height = 0, sc.cam["width"], sc.cam["height"]
    c.Rcl[:] = sc.Rcl.ravel().tolist(); c.Pcl[:] = sc.Pcl.tolist(); c.extR[:] = sc.extR.ravel().tolist(); c.extT[:] = sc.extT.tolist()
    c.img_point_cov = float(sc.cfg["img_point_cov"])
    c.patch_pyrimid_level = int(sc.cfg["patch_pyrimid_level"])
    c.max_iterations = int(max_iterations or sc.cfg["max_iterations"])
    c.exposure_estimate_en, c.inverse_composition_en, c.mp_proc_num = int(exposure), int(inverse), int(mp_proc_num)
    if distortion is not None:                      # radial-tangential d0..d4 of vk::PinholeCamera
        c.cam.distortion = 1
        c.cam.d[:] = [float(x) for x in distortion]
    return c


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    den = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (den if den > 0 else 1.0)


def state_diff(sa, sb):
    """max relative differences between two state structs (rotation as Frobenius, rest as vector norms)."""
    A, B = orc.state_arrays(sa), orc.state_arrays(sb)
    return dict(R=np.linalg.norm(A["R"] - B["R"]), t=relerr(A["t"], B["t"]), P=relerr(A["P"], B["P"]), inv_expo=abs(A["inv_expo"] - B["inv_expo"]),
                rest=max(np.linalg.norm(A[k] - B[k]) for k in ("vel", "bg", "ba", "grav")))
