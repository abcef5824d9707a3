"""Builders of the C-ABI configuration structs from a scenario / parameter dictionaries: the values a LIVMapper would read from its yaml
(reference config/avia.yaml: lio/*, vio/*, extrin_calib/*; camera yaml) laid into `livo2_lidar_cfg` / `livo2_visual_cfg`.
Product-side plumbing: used by bench.py, tools/ and the tests alike; imports nothing from tests/ or oracle/."""
import importlib

DEG2RAD_PCL = 0.017453293       # PCL's DEG2RAD macro (pcl/pcl_macros.h), the constant reference src/voxel_map.cpp:21 compiles against


def _pkg():
    return importlib.import_module(__package__)


def lidar_cfg(sc, max_iterations=None):
    """sc: object with .cfg (max_iterations, max_layer, sigma_num, dept_err, beam_err, voxel_size), .extR (3x3), .extT (3)"""
    c = _pkg().LidarCfg()
    c.max_iterations = int(max_iterations or sc.cfg["max_iterations"])
    c.max_layer = int(sc.cfg["max_layer"])
    c.sigma_num, c.dept_err, c.beam_err, c.voxel_size, c.deg2rad = float(sc.cfg["sigma_num"]), float(sc.cfg["dept_err"]), float(sc.cfg["beam_err"]), float(sc.cfg["voxel_size"]), DEG2RAD_PCL
    c.extR[:] = sc.extR.ravel().tolist()
    c.extT[:] = sc.extT.tolist()
    return c


def visual_cfg(sc, exposure=True, max_iterations=None, inverse=False, mp_proc_num=1, distortion=None, equidistant=None):
    """sc: object with .cam (fx, fy, cx, cy, width, height), .cfg (img_point_cov, patch_pyrimid_level, max_iterations), .Rcl, .Pcl, .extR, .extT.
    distortion: the five radial-tangential coefficients d0..d4 of vk::PinholeCamera; equidistant: k1..k4 of vk::EquidistantCamera."""
    c = _pkg().VisualCfg()
    c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = sc.cam["fx"], sc.cam["fy"], sc.cam["cx"], sc.cam["cy"]
    c.cam.distortion, c.cam.width, c.cam.height = 0, sc.cam["width"], sc.cam["height"]
    c.Rcl[:] = sc.Rcl.ravel().tolist(); c.Pcl[:] = sc.Pcl.tolist(); c.extR[:] = sc.extR.ravel().tolist(); c.extT[:] = sc.extT.tolist()
    c.img_point_cov = float(sc.cfg["img_point_cov"])
    c.patch_pyrimid_level = int(sc.cfg["patch_pyrimid_level"])
    c.max_iterations = int(max_iterations or sc.cfg["max_iterations"])
    c.exposure_estimate_en, c.inverse_composition_en, c.mp_proc_num = int(exposure), int(inverse), int(mp_proc_num)
    if distortion is not None:
        c.cam.distortion = 1
        c.cam.d[:] = [float(x) for x in distortion]
    if equidistant is not None:
        c.cam.distortion = 2
        c.cam.d[:] = [float(x) for x in equidistant] + [0.0]
    return c


def prior_states(sc):
    """(iterate, prior) both at the scenario's prior pose — what StateEstimation / computeJacobianAndUpdateEKF start from"""
    State = _pkg().State
    mk = lambda: State.from_pose(sc.R_prior, sc.t_prior, sc.P, inv_expo=getattr(sc, "tau_prior", 1.0))
    return mk(), mk()
