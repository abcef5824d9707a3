"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes front-end of oracle/liboracle.so (the CPU restatement of the reference path, see orc_api.cpp).
May be imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by the product.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_IT = 16


class StatePOD(C.Structure):
    _fields_ = [("rot", C.c_double * 9), ("pos", C.c_double * 3), ("inv_expo", C.c_double), ("vel", C.c_double * 3), ("bg", C.c_double * 3),
                ("ba", C.c_double * 3), ("grav", C.c_double * 3), ("cov", C.c_double * 361)]


class LidarCfg(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("max_layer", C.c_int), ("sigma_num", C.c_double), ("dept_err", C.c_double), ("beam_err", C.c_double),
                ("voxel_size", C.c_double), ("deg2rad", C.c_double), ("extR", C.c_double * 9), ("extT", C.c_double * 3), ("num_threads", C.c_int),
                ("pad", C.c_int)]


class VisualCfg(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("d", C.c_double * 5), ("distortion", C.c_int),
                ("width", C.c_int), ("height", C.c_int), ("patch_pyrimid_level", C.c_int), ("max_iterations", C.c_int),
                ("exposure_estimate_en", C.c_int), ("inverse_composition_en", C.c_int), ("num_threads", C.c_int), ("img_point_cov", C.c_double),
                ("Rcl", C.c_double * 9), ("Pcl", C.c_double * 3), ("extR", C.c_double * 9), ("extT", C.c_double * 3)]


class LidarIterTrace(C.Structure):
    _fields_ = [("n_eff", C.c_int), ("total_residual", C.c_double), ("HtH", C.c_double * 36), ("Htz", C.c_double * 6), ("solution", C.c_double * 19),
                ("converged", C.c_int), ("stopped", C.c_int)]


class VisualIterTrace(C.Structure):
    _fields_ = [("level", C.c_int), ("iteration", C.c_int), ("accepted", C.c_int), ("n_meas", C.c_int), ("error", C.c_float), ("HtH", C.c_double * 49),
                ("Htz", C.c_double * 7), ("solution", C.c_double * 19)]


def build(kind="golden", out_dir=None):
    """Compile the oracle with the committed Makefile (g++ only).  kind: 'golden' | 'fast'."""
    out_dir = out_dir or _HERE
    subprocess.run(["make", "-C", _HERE, kind, f"OUT={out_dir}"], check=True, capture_output=True)
    return os.path.join(out_dir, "liboracle.so" if kind == "golden" else "liboracle_fast.so")


_libs = {}


def load(path=None):
    path = path or os.path.join(_HERE, "liboracle.so")
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        build("golden")
    lib = C.CDLL(path)
    assert lib.orc_sizeof_state() == C.sizeof(StatePOD)
    assert lib.orc_sizeof_lidar_trace() == C.sizeof(LidarIterTrace), (lib.orc_sizeof_lidar_trace(), C.sizeof(LidarIterTrace))
    assert lib.orc_sizeof_visual_trace() == C.sizeof(VisualIterTrace), (lib.orc_sizeof_visual_trace(), C.sizeof(VisualIterTrace))
    lib.orc_map_create.restype = C.c_void_p
    lib.orc_map_from_flat.restype = C.c_void_p
    _libs[path] = lib
    return lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def make_state(R, t, P, inv_expo=1.0, vel=None, bg=None, ba=None, grav=None, cls=StatePOD):
    s = cls()
    s.rot[:] = np.asarray(R, float).ravel().tolist()
    s.pos[:] = np.asarray(t, float).tolist()
    s.inv_expo = float(inv_expo)
    for name, v in (("vel", vel), ("bg", bg), ("ba", ba), ("grav", grav)):
        getattr(s, name)[:] = (np.zeros(3) if v is None else np.asarray(v, float)).tolist()
    s.cov[:] = np.asarray(P, float).ravel().tolist()
    return s


def state_arrays(s):
    return dict(R=np.array(s.rot).reshape(3, 3), t=np.array(s.pos), inv_expo=s.inv_expo, vel=np.array(s.vel), bg=np.array(s.bg), ba=np.array(s.ba),
                grav=np.array(s.grav), P=np.array(s.cov).reshape(19, 19))


class OracleMap:
    def __init__(self, lib, handle):
        self.lib, self.h = lib, C.c_void_p(handle)

    @classmethod
    def from_flat(cls, fm, lib=None):
        lib = lib or load()
        a = dict(keys=np.ascontiguousarray(fm.root_key, np.int64), rn=np.ascontiguousarray(fm.root_node, np.int32),
                 rc=np.ascontiguousarray(fm.root_center, np.float64), rq=np.ascontiguousarray(fm.root_quarter, np.float32),
                 npl=np.ascontiguousarray(fm.node_plane, np.int32), nch=np.ascontiguousarray(fm.node_child, np.int32),
                 pn=np.ascontiguousarray(fm.plane_normal, np.float64), pc=np.ascontiguousarray(fm.plane_center, np.float64),
                 pv=np.ascontiguousarray(fm.plane_var, np.float64), pd=np.ascontiguousarray(fm.plane_d, np.float32),
                 pr=np.ascontiguousarray(fm.plane_radius, np.float32))
        h = lib.orc_map_from_flat(C.c_double(fm.voxel_size), C.c_int(fm.max_layer), C.c_int(len(a["rn"])), _p(a["keys"], C.c_int64), _p(a["rn"], C.c_int32),
                                  _p(a["rc"], C.c_double), _p(a["rq"], C.c_float), _p(a["npl"], C.c_int32), _p(a["nch"], C.c_int32), _p(a["pn"], C.c_double),
                                  _p(a["pc"], C.c_double), _p(a["pv"], C.c_double), _p(a["pd"], C.c_float), _p(a["pr"], C.c_float))
        return cls(lib, h)

    @classmethod
    def build(cls, pw, var, voxel_size, max_layer, layer_init_num, max_points_num, planer_threshold, lib=None):
        """Restated BuildVoxelMap (reference src/voxel_map.cpp:532-591)."""
        lib = lib or load()
        lin = (C.c_int * 5)(*list(layer_init_num)[:5])
        h = lib.orc_map_create(C.c_double(voxel_size), C.c_int(max_layer), lin, C.c_int(max_points_num), C.c_double(planer_threshold))
        m = cls(lib, h)
        pw = np.ascontiguousarray(pw, np.float64); var = np.ascontiguousarray(var, np.float64)
        lib.orc_map_build(m.h, _p(pw, C.c_double), _p(var, C.c_double), C.c_int(len(pw)))
        return m

    def update(self, pw, var):
        pw = np.ascontiguousarray(pw, np.float64); var = np.ascontiguousarray(var, np.float64)
        self.lib.orc_map_update(self.h, _p(pw, C.c_double), _p(var, C.c_double), C.c_int(len(pw)))

    def slide(self, position_last, sliding_thresh, half_map_size):
        """Restated mapSliding / clearMemOutOfMap (reference src/voxel_map.cpp:924-972): root voxels deleted, -1 below the threshold."""
        pos = np.ascontiguousarray(position_last, np.float64).reshape(3)
        self.lib.orc_map_slide.restype = C.c_int
        return int(self.lib.orc_map_slide(self.h, _p(pos, C.c_double), C.c_double(sliding_thresh), C.c_int(half_map_size)))

    def export(self, voxel_size, max_layer):
        from scenarios.synth import FlatMap
        nr, nn, npl = C.c_int(), C.c_int(), C.c_int()
        self.lib.orc_map_counts(self.h, C.byref(nr), C.byref(nn), C.byref(npl))
        R, N, P = nr.value, nn.value, npl.value
        fm = FlatMap(voxel_size, max_layer, np.zeros((R, 3), np.int64), np.zeros(R, np.int32), np.zeros((R, 3)), np.zeros(R, np.float32),
                     np.zeros(N, np.int32), np.zeros((N, 8), np.int32), np.zeros((P, 3)), np.zeros((P, 3)), np.zeros((P, 36)), np.zeros(P, np.float32),
                     np.zeros(P, np.float32))
        self.lib.orc_map_export(self.h, _p(fm.root_key, C.c_int64), _p(fm.root_node, C.c_int32), _p(fm.root_center, C.c_double), _p(fm.root_quarter, C.c_float),
                                _p(fm.node_plane, C.c_int32), _p(fm.node_child, C.c_int32), _p(fm.plane_normal, C.c_double), _p(fm.plane_center, C.c_double),
                                _p(fm.plane_var, C.c_double), _p(fm.plane_d, C.c_float), _p(fm.plane_radius, C.c_float))
        return fm

    def __del__(self):
        try:
            if self.h:
                self.lib.orc_map_destroy(self.h)
                self.h = None
        except Exception:
            pass


class PlaneFit(C.Structure):
    """VoxelPlane fields written by VoxelOctoTree::init_plane (include/voxel_map.h:69-94, src/voxel_map.cpp:55-135)."""
    _fields_ = [("center", C.c_double * 3), ("normal", C.c_double * 3), ("y_normal", C.c_double * 3), ("x_normal", C.c_double * 3),
                ("covariance", C.c_double * 9), ("plane_var", C.c_double * 36), ("radius", C.c_float), ("min_eigen_value", C.c_float),
                ("mid_eigen_value", C.c_float), ("max_eigen_value", C.c_float), ("d", C.c_float), ("points_size", C.c_int32),
                ("is_plane", C.c_int32), ("pad", C.c_int32)]


def init_plane(point_w, var, planer_threshold, lib=None):
    lib = lib or load()
    pw = np.ascontiguousarray(point_w, np.float64).reshape(-1, 3)
    v = np.ascontiguousarray(var, np.float64).reshape(-1, 9)
    out = PlaneFit()
    lib.orc_init_plane.restype = C.c_int
    lib.orc_init_plane.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_float, C.POINTER(PlaneFit)]
    lib.orc_init_plane(_p(pw, C.c_double), _p(v, C.c_double), len(pw), float(planer_threshold), C.byref(out))
    return out


def init_plane_batch(point_w, var, offsets, planer_threshold, lib=None):
    """returns (array of PlaneFit, seconds spent inside the C++ loop)"""
    lib = lib or load()
    pw = np.ascontiguousarray(point_w, np.float64).reshape(-1, 3)
    v = np.ascontiguousarray(var, np.float64).reshape(-1, 9)
    off = np.ascontiguousarray(offsets, np.int32)
    out = (PlaneFit * (len(off) - 1))()
    lib.orc_init_plane_batch.restype = C.c_double
    lib.orc_init_plane_batch.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_int, C.c_float, C.POINTER(PlaneFit)]
    secs = lib.orc_init_plane_batch(_p(pw, C.c_double), _p(v, C.c_double), _p(off, C.c_int), len(off) - 1, float(planer_threshold), out)
    return out, secs


class WarpCfg(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("width", C.c_int32), ("height", C.c_int32),
                ("R_cur", C.c_double * 9), ("t_cur", C.c_double * 3), ("inv_expo_cur", C.c_double), ("patch_pyrimid_level", C.c_int32),
                ("normal_en", C.c_int32), ("ncc_en", C.c_int32), ("pad", C.c_int32), ("ncc_thre", C.c_double), ("outlier_threshold", C.c_double),
                ("d", C.c_double * 5), ("distortion", C.c_int32), ("pad2", C.c_int32)]


def _cam_distortion(c, cam):
    """cam["d"] (optional): the five radial-tangential coefficients d0..d4 of vk::PinholeCamera; cam["k"] (optional): k1..k4 of vk::EquidistantCamera"""
    d, k = cam.get("d"), cam.get("k")
    c.distortion = 2 if k is not None else (0 if d is None else 1)
    c.d[:] = ([float(x) for x in k] + [0.0]) if k is not None else ([0.0] * 5 if d is None else [float(x) for x in d])


def cam_roundtrip(cam, uv, lib=None):
    """vk::PinholeCamera::cam2world then world2cam for pixels uv [n,2] of the camera dict `cam` (optional key "d": radial-tangential coefficients).
    Returns (bearing vectors [n,3], re-projected pixels [n,2])."""
    lib = lib or load()
    c = WarpCfg()
    c.fx, c.fy, c.cx, c.cy, c.width, c.height = cam["fx"], cam["fy"], cam["cx"], cam["cy"], cam["width"], cam["height"]
    _cam_distortion(c, cam)
    uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
    f, px = np.zeros((len(uv), 3)), np.zeros((len(uv), 2))
    lib.orc_cam2world.restype = None; lib.orc_world2cam.restype = None
    for i in range(len(uv)):
        lib.orc_cam2world(C.byref(c), C.c_double(uv[i, 0]), C.c_double(uv[i, 1]), f[i].ctypes.data_as(C.POINTER(C.c_double)))
        lib.orc_world2cam(C.byref(c), f[i].ctypes.data_as(C.POINTER(C.c_double)), px[i].ctypes.data_as(C.POINTER(C.c_double)))
    return f, px


def warp_candidates(rs, lib=None):
    """Per-point tail of retrieveFromVisualSparseMap over a scenarios.synth.RetrieveScenario; returns a dict of per-candidate arrays.
    rs.ref_id (optional attribute) = ref_ftr->id_ per candidate: the warp_map reuse of the !normal_en branch (src/vio.cpp:716-734)."""
    lib = lib or load()
    c = WarpCfg()
    c.fx, c.fy, c.cx, c.cy, c.width, c.height = rs.cam["fx"], rs.cam["fy"], rs.cam["cx"], rs.cam["cy"], rs.cam["width"], rs.cam["height"]
    _cam_distortion(c, rs.cam)
    c.R_cur[:] = rs.R_cur.ravel().tolist(); c.t_cur[:] = rs.t_cur.tolist(); c.inv_expo_cur = rs.inv_expo_cur
    c.patch_pyrimid_level, c.normal_en, c.ncc_en = int(rs.cfg["patch_pyrimid_level"]), int(rs.cfg["normal_en"]), int(rs.cfg["ncc_en"])
    c.ncc_thre, c.outlier_threshold = float(rs.cfg["ncc_thre"]), float(rs.cfg["outlier_threshold"])
    n, L = len(rs.pos), int(rs.cfg["patch_pyrimid_level"])
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    pos, normal, px, f, R, t, ie = f64(rs.pos), f64(rs.normal), f64(rs.ref_px), f64(rs.ref_f), f64(rs.ref_R), f64(rs.ref_t), f64(rs.ref_inv_expo)
    idx, lvl = np.ascontiguousarray(rs.ref_img_idx, np.int32), np.ascontiguousarray(rs.ref_level, np.int32)
    img, refs = np.ascontiguousarray(rs.img, np.uint8), np.ascontiguousarray(rs.ref_imgs, np.uint8)
    out = dict(accepted=np.zeros(n, np.int32), search_level=np.zeros(n, np.int32), error=np.zeros(n, np.float32), ncc=np.zeros(n), A=np.zeros((n, 4)),
               patch_wrap=np.zeros((n, L, 64), np.float32))
    lib.orc_warp_candidates.restype = C.c_double
    lib.orc_warp_candidates.argtypes = [C.POINTER(WarpCfg), C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_int] + [C.c_void_p] * 16
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    rid = getattr(rs, "ref_id", None)
    rid = None if rid is None else np.ascontiguousarray(rid, np.int32)
    out["seconds"] = lib.orc_warp_candidates(C.byref(c), _p(img, C.c_uint8), _p(refs, C.c_uint8), n, vp(pos), vp(normal), vp(idx), vp(px), vp(f), vp(R), vp(t),
                                             vp(lvl), vp(ie), None if rid is None else vp(rid), vp(out["accepted"]), vp(out["search_level"]), vp(out["error"]),
                                             vp(out["ncc"]), vp(out["A"]), vp(out["patch_wrap"]))
    return out


def undistort(xyz, curvature, poses, rot_end, pos_end, extR, extT, lib=None):
    """ImuProcess::UndistortPcl backward loop (src/IMU_Processing.cpp:494-539) on a copy of xyz; poses: [n_poses][22] (Pose6D rows)."""
    lib = lib or load()
    out = np.ascontiguousarray(xyz, np.float32).copy()
    cur = np.ascontiguousarray(curvature, np.float32)
    P = np.ascontiguousarray(poses, np.float64).reshape(-1, 22)
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    a, b, c, d = f64(rot_end), f64(pos_end), f64(extR), f64(extT)
    lib.orc_undistort.restype = None
    lib.orc_undistort.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    vp = lambda x: x.ctypes.data_as(C.c_void_p)
    lib.orc_undistort(vp(out), vp(cur), len(out), vp(P), len(P), vp(a), vp(b), vp(c), vp(d))
    return out


def voxel_grid(xyz, leaf, lib=None):
    """pcl::VoxelGrid centroid filter as restated in orc_preprocess.hpp; returns [m][3] float32 (ascending leaf index)."""
    lib = lib or load()
    x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
    out = np.zeros_like(x)
    lib.orc_voxel_grid.restype = C.c_int
    lib.orc_voxel_grid.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
    m = lib.orc_voxel_grid(x.ctypes.data_as(C.c_void_p), len(x), float(leaf), out.ctypes.data_as(C.c_void_p))
    if m < 0:
        raise OverflowError("leaf grid overflows int32")
    return out[:m].copy()


class SelectCfg(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("width", C.c_int32), ("height", C.c_int32),
                ("R_cur", C.c_double * 9), ("t_cur", C.c_double * 3), ("border", C.c_int32), ("grid_size", C.c_int32), ("grid_n_width", C.c_int32),
                ("grid_n_height", C.c_int32), ("patch_size_half", C.c_int32), ("pad", C.c_int32), ("d", C.c_double * 5), ("distortion", C.c_int32), ("raycast_en", C.c_int32)]


def visual_select(ss, lib=None, raycast=False, omap=None):
    """Selection half of retrieveFromVisualSparseMap over a scenarios.synth.SelectScenario; returns per-cell / per-point arrays.  raycast: the RayCasting module
    (vio.cpp:487-591) with `omap` (OracleMap or None) as plane_map; adds `add_from_voxel_map` [k][6] = center_, normal_."""
    lib = lib or load()
    c = SelectCfg()
    c.raycast_en = 1 if raycast else 0
    c.fx, c.fy, c.cx, c.cy, c.width, c.height = ss.cam["fx"], ss.cam["fy"], ss.cam["cx"], ss.cam["cy"], ss.cam["width"], ss.cam["height"]
    _cam_distortion(c, ss.cam)
    c.R_cur[:] = ss.R_cur.ravel().tolist(); c.t_cur[:] = ss.t_cur.tolist()
    c.border, c.grid_size, c.grid_n_width, c.grid_n_height, c.patch_size_half = ss.border, ss.grid_size, ss.grid_n_width, ss.grid_n_height, 4
    length = ss.grid_n_width * ss.grid_n_height
    pg = np.ascontiguousarray(ss.pg, np.float64); pos = np.ascontiguousarray(ss.pos, np.float64)
    keys = np.ascontiguousarray(ss.keys, np.int64); act = np.ascontiguousarray(ss.active, np.uint8)
    out = dict(cell_point=np.zeros(length, np.int32), cell_dist=np.zeros(length, np.float32), cell_type=np.zeros(length, np.int32), discont=np.zeros(length, np.int32),
               in_fov=np.zeros(len(pos), np.int32), depth_img=np.zeros((ss.cam["height"], ss.cam["width"]), np.float32))
    lib.orc_visual_select_rc.restype = C.c_double
    lib.orc_visual_select_rc.argtypes = [C.POINTER(SelectCfg), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    add = np.zeros((length, 6)); n_add = C.c_int32(0)
    out["seconds"] = lib.orc_visual_select_rc(C.byref(c), vp(pg), len(pg), vp(pos), vp(keys), vp(act), len(pos), vp(out["cell_point"]), vp(out["cell_dist"]),
                                              vp(out["cell_type"]), vp(out["discont"]), vp(out["in_fov"]), vp(out["depth_img"]), omap.h if omap is not None else None,
                                              vp(add), length, C.cast(C.byref(n_add), C.c_void_p))
    out["add_from_voxel_map"] = add[:n_add.value]
    return out


def choose_ref(cs, cell_point, cell_discont, ref_patch=None, lib=None):
    """Reference-patch choice (src/vio.cpp:644-696) over a scenarios.synth.RetrieveChainScenario for the cells of one selection.
    Returns (cell_obs [length] int32: global observation index of ref_ftr or -1, ref_patch [n] int32 after the call)."""
    lib = lib or load()
    f64 = lambda a: np.ascontiguousarray(a, np.float64)
    cp, cd = np.ascontiguousarray(cell_point, np.int32), np.ascontiguousarray(cell_discont, np.int32)
    rp = np.ascontiguousarray(cs.ref_patch if ref_patch is None else ref_patch, np.int32).copy()
    R, t, pos = f64(cs.sel.R_cur), f64(cs.sel.t_cur), f64(cs.sel.pos)
    off, oid = np.ascontiguousarray(cs.obs_offset, np.int32), np.ascontiguousarray(cs.obs_id, np.int32)
    oR, ot, op = f64(cs.obs_R), f64(cs.obs_t), np.ascontiguousarray(cs.obs_patch, np.float32)
    ni = np.ascontiguousarray(cs.normal_initialized, np.uint8)
    out = np.zeros(len(cp), np.int32)
    lib.orc_choose_ref.restype = None
    lib.orc_choose_ref.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 11
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.orc_choose_ref(int(cs.cfg["normal_en"]), vp(R), vp(t), len(cp), vp(cp), vp(cd), vp(pos), vp(off), vp(oid), vp(oR), vp(ot), vp(op), vp(ni), vp(rp), vp(out))
    return out, rp


class _Cand:
    pass


def visual_retrieve(cs, lib=None, raycast=False, omap=None):
    """The whole retrieveFromVisualSparseMap (raycast_en = `raycast`, plane_map = `omap`) over a RetrieveChainScenario: selection -> reference-patch choice -> warp / gate
    tail.  Returns the stage outputs: sel (dict of visual_select), cell_obs, ref_patch, cand_cell (grid cell of every candidate, ascending),
    cand_point, cand_obs, tail (dict of warp_candidates), and the appended sub-map: sub_point, sub_obs (survivors in order)."""
    lib = lib or load()
    sel = visual_select(cs.sel, lib, raycast=raycast, omap=omap)
    cell_obs, ref_patch = choose_ref(cs, sel["cell_point"], sel["discont"], lib=lib)
    cand_cell = np.nonzero(cell_obs >= 0)[0].astype(np.int32)
    cand_point, cand_obs = sel["cell_point"][cand_cell], cell_obs[cand_cell]
    r = _Cand()
    r.cam, r.cfg, r.img, r.ref_imgs, r.R_cur, r.t_cur, r.inv_expo_cur = cs.sel.cam, cs.cfg, cs.img, cs.ref_imgs, cs.sel.R_cur, cs.sel.t_cur, cs.inv_expo_cur
    r.pos, r.normal = cs.sel.pos[cand_point], cs.normal[cand_point]
    r.ref_img_idx, r.ref_px, r.ref_f, r.ref_R, r.ref_t = cs.obs_img_idx[cand_obs], cs.obs_px[cand_obs], cs.obs_f[cand_obs], cs.obs_R[cand_obs], cs.obs_t[cand_obs]
    r.ref_level, r.ref_inv_expo, r.ref_id = cs.obs_level[cand_obs], cs.obs_inv_expo[cand_obs], cs.obs_id[cand_obs]
    tail = warp_candidates(r, lib)
    keep = tail["accepted"] != 0
    return dict(sel=sel, cell_obs=cell_obs, ref_patch=ref_patch, cand_cell=cand_cell, cand_point=cand_point, cand_obs=cand_obs, tail=tail,
                sub_point=cand_point[keep], sub_obs=cand_obs[keep])


def feat_map_keys(pos, lib=None):
    lib = lib or load()
    pos = np.ascontiguousarray(pos, np.float64).reshape(-1, 3)
    keys = np.zeros((len(pos), 3), np.int64)
    lib.orc_feat_map_key.restype = None
    lib.orc_feat_map_key.argtypes = [C.c_void_p, C.c_void_p]
    for i in range(len(pos)):
        lib.orc_feat_map_key(pos[i].ctypes.data_as(C.c_void_p), keys[i].ctypes.data_as(C.c_void_p))
    return keys


class ImuCfg(C.Structure):
    _fields_ = [("cov_gyr", C.c_double * 3), ("cov_acc", C.c_double * 3), ("cov_bias_gyr", C.c_double * 3), ("cov_bias_acc", C.c_double * 3), ("cov_inv_expo", C.c_double),
                ("G_m_s2", C.c_double), ("mean_acc_norm", C.c_double), ("ba_bg_est_en", C.c_int32), ("gravity_est_en", C.c_int32), ("exposure_estimate_en", C.c_int32),
                ("first_call", C.c_int32)]


def imu_cfg(d, cls=ImuCfg):
    c = cls()
    for k in ("cov_gyr", "cov_acc", "cov_bias_gyr", "cov_bias_acc"):
        getattr(c, k)[:] = [float(x) for x in d[k]]
    c.cov_inv_expo, c.G_m_s2, c.mean_acc_norm = float(d["cov_inv_expo"]), float(d["G_m_s2"]), float(d["mean_acc_norm"])
    c.ba_bg_est_en, c.gravity_est_en, c.exposure_estimate_en = int(d["ba_bg_est_en"]), int(d["gravity_est_en"]), int(d["exposure_estimate_en"])
    c.first_call = int(d.get("first_call", 0))
    return c


def imu_propagate(state_in, steps, cfg_dict, lib=None):
    """IMU forward propagation (src/IMU_Processing.cpp:298-445); steps: [n][8] = gyr3, acc3, dt, offs_t.  Returns (state_out, poses [n][22], seconds)."""
    lib = lib or load()
    S = np.ascontiguousarray(steps, np.float64).reshape(-1, 8)
    out = StatePOD()
    poses = np.zeros((max(len(S), 1), 22))
    c = imu_cfg(cfg_dict)
    lib.orc_imu_propagate.restype = C.c_double
    lib.orc_imu_propagate.argtypes = [C.POINTER(StatePOD), C.c_void_p, C.c_int, C.POINTER(ImuCfg), C.POINTER(StatePOD), C.c_void_p]
    secs = lib.orc_imu_propagate(C.byref(state_in), S.ctypes.data_as(C.c_void_p), len(S), C.byref(c), C.byref(out), poses.ctypes.data_as(C.c_void_p))
    return out, poses[: len(S)], secs


def lidar_cfg(c, extR, extT, num_threads=1, deg2rad=0.017453293):
    cfg = LidarCfg()
    cfg.max_iterations, cfg.max_layer = int(c["max_iterations"]), int(c["max_layer"])
    cfg.sigma_num, cfg.dept_err, cfg.beam_err, cfg.voxel_size, cfg.deg2rad = float(c["sigma_num"]), float(c["dept_err"]), float(c["beam_err"]), float(c["voxel_size"]), deg2rad
    cfg.extR[:] = np.asarray(extR, float).ravel().tolist()
    cfg.extT[:] = np.asarray(extT, float).tolist()
    cfg.num_threads = num_threads
    return cfg


def _point_bufs(n):
    return dict(match_plane=np.zeros(n, np.int32), dis=np.zeros(n, np.float32), pw=np.zeros((n, 3), np.float32), var=np.zeros((n, 9)),
                body_cov=np.zeros((n, 9)), cross_mat=np.zeros((n, 9)), normal=np.zeros((n, 3)), Rinv=np.zeros(n), Hrow=np.zeros((n, 6)))


def _point_args(b):
    return [_p(b["match_plane"], C.c_int32), _p(b["dis"], C.c_float), _p(b["pw"], C.c_float), _p(b["var"], C.c_double), _p(b["body_cov"], C.c_double),
            _p(b["cross_mat"], C.c_double), _p(b["normal"], C.c_double), _p(b["Rinv"], C.c_double), _p(b["Hrow"], C.c_double)]


def lidar_iterate(omap, cfg, xyz, cur, prop):
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = len(xyz)
    HtH, Htz, neff, tr = np.zeros(36), np.zeros(6), C.c_int(), C.c_double()
    b = _point_bufs(n)
    omap.lib.orc_lidar_iterate(omap.h, C.byref(cfg), _p(xyz, C.c_float), C.c_int(n), C.byref(cur), C.byref(prop), _p(HtH, C.c_double), _p(Htz, C.c_double),
                               C.byref(neff), C.byref(tr), *_point_args(b))
    return dict(HtH=HtH.reshape(6, 6), Htz=Htz, n_eff=neff.value, total_residual=tr.value, **b)


def lidar_state_estimation(omap, cfg, xyz, state_in, prop, want_points=True):
    xyz = np.ascontiguousarray(xyz, np.float32)
    n = len(xyz)
    out = StatePOD()
    trace = (LidarIterTrace * MAX_IT)()
    nit, secs = C.c_int(), C.c_double()
    b = _point_bufs(n if want_points else 0)
    args = _point_args(b) if want_points else [None] * 9
    omap.lib.orc_lidar_state_estimation(omap.h, C.byref(cfg), _p(xyz, C.c_float), C.c_int(n), C.byref(state_in), C.byref(prop), C.byref(out),
                                        C.cast(trace, C.c_void_p), C.byref(nit), C.byref(secs), *args)
    return dict(state=out, n_iters=nit.value, seconds=secs.value, trace=[trace[i] for i in range(nit.value)], **(b if want_points else {}))


def visual_cfg(sc, num_threads=1, exposure=True, inverse=False, max_iterations=None, distortion=None, equidistant=None):
    cfg = VisualCfg()
    cfg.fx, cfg.fy, cfg.cx, cfg.cy = sc.cam["fx"], sc.cam["fy"], sc.cam["cx"], sc.cam["cy"]
    cfg.distortion, cfg.width, cfg.height = 0, sc.cam["width"], sc.cam["height"]
    if distortion is not None:                      # vk::PinholeCamera radial-tangential d0..d4
        cfg.distortion = 1
        cfg.d[:] = [float(x) for x in distortion]
    if equidistant is not None:                     # vk::EquidistantCamera k1..k4
        cfg.distortion = 2
        cfg.d[:] = [float(x) for x in equidistant] + [0.0]
    cfg.patch_pyrimid_level = int(sc.cfg["patch_pyrimid_level"])
    cfg.max_iterations = int(max_iterations or sc.cfg["max_iterations"])
    cfg.exposure_estimate_en, cfg.inverse_composition_en, cfg.num_threads = int(exposure), int(inverse), num_threads
    cfg.img_point_cov = float(sc.cfg["img_point_cov"])
    cfg.Rcl[:] = sc.Rcl.ravel().tolist(); cfg.Pcl[:] = sc.Pcl.tolist(); cfg.extR[:] = sc.extR.ravel().tolist(); cfg.extT[:] = sc.extT.tolist()
    return cfg


def visual_iterate(cfg, sc, level, cur, lib=None):
    lib = lib or load()
    M = len(sc.pos)
    img = np.ascontiguousarray(sc.img, np.uint8); pos = np.ascontiguousarray(sc.pos, np.float64); wp = np.ascontiguousarray(sc.warp_patch, np.float32)
    sl = np.ascontiguousarray(sc.search_levels, np.int32); ie = np.ascontiguousarray(sc.inv_expo_list, np.float64)
    z, H, err = np.zeros(M * 64), np.zeros((M * 64, 7)), np.zeros(M, np.float32)
    HtH, Htz, e, nm = np.zeros(49), np.zeros(7), C.c_float(), C.c_int()
    lib.orc_visual_iterate(C.byref(cfg), _p(img, C.c_uint8), _p(pos, C.c_double), _p(wp, C.c_float), _p(sl, C.c_int32), _p(ie, C.c_double), C.c_int(M),
                           C.c_int(level), C.byref(cur), _p(z, C.c_double), _p(H, C.c_double), _p(err, C.c_float), _p(HtH, C.c_double), _p(Htz, C.c_double),
                           C.byref(e), C.byref(nm))
    return dict(z=z, H=H, errors=err, HtH=HtH.reshape(7, 7), Htz=Htz, error=e.value, n_meas=nm.value)


def _ref_args(sc):
    ri = np.ascontiguousarray(sc.ref_imgs, np.uint8); idx = np.ascontiguousarray(sc.ref_img_idx, np.int32)
    px = np.ascontiguousarray(sc.ref_px, np.float64); f = np.ascontiguousarray(sc.ref_f, np.float64)
    R = np.ascontiguousarray(sc.ref_R, np.float64); pos = np.ascontiguousarray(sc.ref_pos, np.float64)
    sc._ref_keep = (ri, idx, px, f, R, pos)
    return [_p(ri, C.c_uint8), _p(idx, C.c_int32), _p(px, C.c_double), _p(f, C.c_double), _p(R, C.c_double), _p(pos, C.c_double)]


def visual_iterate_inverse(cfg, sc, level, cur, lib=None):
    lib = lib or load()
    M = len(sc.pos)
    img = np.ascontiguousarray(sc.img, np.uint8); pos = np.ascontiguousarray(sc.pos, np.float64); wp = np.ascontiguousarray(sc.warp_patch, np.float32)
    sl = np.ascontiguousarray(sc.search_levels, np.int32); ie = np.ascontiguousarray(sc.inv_expo_list, np.float64)
    z, H, err = np.zeros(M * 64), np.zeros((M * 64, 6)), np.zeros(M, np.float32)
    HtH, Htz, e, nm = np.zeros(36), np.zeros(6), C.c_float(), C.c_int()
    lib.orc_visual_iterate_inverse(C.byref(cfg), _p(img, C.c_uint8), _p(pos, C.c_double), _p(wp, C.c_float), _p(sl, C.c_int32), _p(ie, C.c_double), C.c_int(M),
                                   C.c_int(level), C.byref(cur), *_ref_args(sc), _p(z, C.c_double), _p(H, C.c_double), _p(err, C.c_float), _p(HtH, C.c_double),
                                   _p(Htz, C.c_double), C.byref(e), C.byref(nm))
    return dict(z=z, H=H, errors=err, HtH=HtH.reshape(6, 6), Htz=Htz, error=e.value, n_meas=nm.value)


def visual_update(cfg, sc, state_in, prop, lib=None):
    lib = lib or load()
    M = len(sc.pos)
    img = np.ascontiguousarray(sc.img, np.uint8); pos = np.ascontiguousarray(sc.pos, np.float64); wp = np.ascontiguousarray(sc.warp_patch, np.float32)
    sl = np.ascontiguousarray(sc.search_levels, np.int32); ie = np.ascontiguousarray(sc.inv_expo_list, np.float64)
    out = StatePOD(); err = np.zeros(M, np.float32)
    trace = (VisualIterTrace * (8 * MAX_IT))()
    nt, secs = C.c_int(), C.c_double()
    G, RP = np.zeros(361), np.zeros(12)
    refs = _ref_args(sc) if cfg.inverse_composition_en else [None] * 6
    lib.orc_visual_update(C.byref(cfg), _p(img, C.c_uint8), _p(pos, C.c_double), _p(wp, C.c_float), _p(sl, C.c_int32), _p(ie, C.c_double), C.c_int(M),
                          C.byref(state_in), C.byref(prop), C.byref(out), _p(err, C.c_float), C.cast(trace, C.c_void_p), C.byref(nt), C.byref(secs),
                          _p(G, C.c_double), _p(RP, C.c_double), *refs)
    return dict(state=out, errors=err, trace=[trace[i] for i in range(nt.value)], seconds=secs.value, G=G.reshape(19, 19), Rcw=RP[:9].reshape(3, 3), Pcw=RP[9:])
