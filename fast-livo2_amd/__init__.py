"""fast-livo2_amd — MI355X (gfx950) ESIKF measurement update of FAST-LIVO2 behind a C ABI.

The product is `lib/liblivo2_hip.so` (sources in `csrc/`, ABI in `../include/livo2_hip.h`) plus the C++ host shim in
`host/` that mirrors `VoxelMapManager::StateEstimation` / `VIOManager::computeJacobianAndUpdateEKF`.
This Python package is a thin numpy/ctypes convenience layer over that ABI used by tests/, bench.py and smoke();
import it with `importlib.import_module("fast-livo2_amd")` (the directory name is not a Python identifier).
"""
import ctypes as C

import numpy as np

from . import abi
from .abi import (Cam, ImuCfg, LidarCfg, MapTreeCfg, LidarPoints, LidarResult, LidarSums, MapView, PlaneFit, RetrieveCandidates, RetrieveCfg, RetrieveChainOut, RetrieveOut, SelectCfg, State, VisualCfg, VisualObs, VisualResult, VisualSums)  # noqa: F401


class Livo2Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"livo2 error {code}: {msg}")
        self.code = code


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _cam_distortion(c, cam):
    """cam["d"] (optional): the five radial-tangential coefficients d0..d4 of vk::PinholeCamera; cam["k"] (optional): k1..k4 of vk::EquidistantCamera"""
    if cam.get("k") is not None:
        c.distortion = 2
        c.d[:] = [float(x) for x in cam["k"]] + [0.0]
        return
    d = cam.get("d")
    c.distortion = 0 if d is None else 1
    c.d[:] = [0.0] * 5 if d is None else [float(x) for x in d]


class Context:
    """One GPU + one HIP stream (livo2_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = abi.load_library()
        h = C.c_void_p()
        if stream is None:
            rc = self.lib.livo2_ctx_create(int(device), C.byref(h))
        else:
            rc = self.lib.livo2_ctx_create_on_stream(int(device), C.c_void_p(int(stream)), C.byref(h))
        if rc != abi.OK:
            raise Livo2Error(rc, "livo2_ctx_create failed (no gfx950 device / HIP runtime?)")
        self.h = h
        self.device = int(device)
        self._keep = {}
        self.n = 0
        self.M = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.livo2_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != abi.OK:
            raise Livo2Error(rc, self.lib.livo2_last_error(self.h).decode())

    def synchronize(self):
        self._chk(self.lib.livo2_ctx_synchronize(self.h))

    @property
    def stream(self):
        return self.lib.livo2_ctx_stream(self.h)

    def set_option(self, name, value):
        self._chk(self.lib.livo2_ctx_set_option(self.h, name.encode(), int(value)))

    def counter(self, name):
        v = C.c_int64()
        self._chk(self.lib.livo2_ctx_get_counter(self.h, name.encode(), C.byref(v)))
        return v.value

    def redzone_check(self):
        """(mode, damaged guard words): LIVO2_REDZONE debug allocator; raises Livo2Error naming the allocation when a guard was overwritten"""
        m, bad = C.c_int32(), C.c_int64()
        self._chk(self.lib.livo2_debug_redzone_check(self.h, C.byref(m), C.byref(bad)))
        return m.value, bad.value

    def kernel_timing(self, enable):
        self._chk(self.lib.livo2_ctx_kernel_timing(self.h, 1 if enable else 0))

    def kernel_timing_read(self, which, reset=True):
        ms, n = C.c_double(), C.c_int64()
        self._chk(self.lib.livo2_ctx_kernel_timing_read(self.h, which, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    # ---- map -------------------------------------------------------------------------------------------------
    def upload_map(self, fm):
        """fm: object with the flat-map arrays (scenarios.FlatMap)."""
        arrs = dict(
            root_key=np.ascontiguousarray(fm.root_key, np.int64), root_node=np.ascontiguousarray(fm.root_node, np.int32),
            root_center=_f64(fm.root_center), root_quarter=np.ascontiguousarray(fm.root_quarter, np.float32),
            node_plane=np.ascontiguousarray(fm.node_plane, np.int32), node_child=np.ascontiguousarray(fm.node_child, np.int32),
            plane_normal=_f64(fm.plane_normal), plane_center=_f64(fm.plane_center), plane_var=_f64(fm.plane_var),
            plane_d=np.ascontiguousarray(fm.plane_d, np.float32), plane_radius=np.ascontiguousarray(fm.plane_radius, np.float32))
        mv = MapView()
        mv.n_roots, mv.n_nodes, mv.n_planes = len(arrs["root_node"]), len(arrs["node_plane"]), len(arrs["plane_d"])
        mv.root_key = abi.as_ptr(arrs["root_key"], C.c_int64)
        mv.root_node = abi.as_ptr(arrs["root_node"], C.c_int32)
        mv.root_center = abi.as_ptr(arrs["root_center"], C.c_double)
        mv.root_quarter = abi.as_ptr(arrs["root_quarter"], C.c_float)
        mv.node_plane = abi.as_ptr(arrs["node_plane"], C.c_int32)
        mv.node_child = abi.as_ptr(arrs["node_child"], C.c_int32)
        mv.plane_normal = abi.as_ptr(arrs["plane_normal"], C.c_double)
        mv.plane_center = abi.as_ptr(arrs["plane_center"], C.c_double)
        mv.plane_var = abi.as_ptr(arrs["plane_var"], C.c_double)
        mv.plane_d = abi.as_ptr(arrs["plane_d"], C.c_float)
        mv.plane_radius = abi.as_ptr(arrs["plane_radius"], C.c_float)
        self._chk(self.lib.livo2_map_upload(self.h, C.byref(mv)))

    def update_planes(self, idx, normal, center, plane_var, d, radius):
        idx = np.ascontiguousarray(idx, np.int32)
        normal, center, plane_var = _f64(normal), _f64(center), _f64(plane_var)
        d, radius = np.ascontiguousarray(d, np.float32), np.ascontiguousarray(radius, np.float32)
        self._chk(self.lib.livo2_map_update_planes(self.h, abi.as_ptr(idx, C.c_int32), len(idx), abi.as_ptr(normal, C.c_double),
                                                   abi.as_ptr(center, C.c_double), abi.as_ptr(plane_var, C.c_double), abi.as_ptr(d, C.c_float),
                                                   abi.as_ptr(radius, C.c_float)))

    # ---- device-resident VoxelMap -----------------------------------------------------------------------------
    def map_tree_create(self, lio_cfg, max_roots, **caps):
        """lio_cfg: dict with voxel_size, min_eigen_value, max_layer, max_points_num, layer_init_num (config/avia.yaml lio/...)."""
        c = MapTreeCfg()
        c.voxel_size, c.planer_threshold = float(lio_cfg["voxel_size"]), float(lio_cfg["min_eigen_value"])
        c.max_layer, c.max_points_num, c.max_roots = int(lio_cfg["max_layer"]), int(lio_cfg["max_points_num"]), int(max_roots)
        c.layer_init_num[:] = [int(v) for v in list(lio_cfg["layer_init_num"])[:5]]
        for k, v in caps.items():
            setattr(c, k, int(v))
        self._chk(self.lib.livo2_map_tree_create(self.h, C.byref(c)))

    def map_tree_update(self, point_w, var, build=False):
        pw, v = _f64(point_w).reshape(-1, 3), _f64(var).reshape(-1, 9)
        self._chk(self.lib.livo2_map_tree_update(self.h, abi.as_ptr(pw, C.c_double), abi.as_ptr(v, C.c_double), len(pw), 1 if build else 0))

    def map_tree_update_from_scan(self, state, cfg, build=False):
        self._chk(self.lib.livo2_map_tree_update_from_scan(self.h, C.byref(state), C.byref(cfg), 1 if build else 0))

    def map_tree_update_from_scan_async(self, state, cfg):
        """UpdateVoxelMap on the context's second stream (state None: the posterior of the last LiDAR update, taken on the device); map_tree_update_join() collects it"""
        self._chk(self.lib.livo2_map_tree_update_from_scan_async(self.h, C.byref(state) if state is not None else None, C.byref(cfg)))

    def map_tree_update_join(self):
        self._chk(self.lib.livo2_map_tree_update_join(self.h))

    def map_tree_stats(self):
        c = np.zeros(8, np.int32)
        self._chk(self.lib.livo2_map_tree_stats(self.h, abi.as_ptr(c, C.c_int32)))
        return dict(nodes=int(c[0]), points=int(c[1]), planes=int(c[2]), cand=int(c[3]), error=int(c[5]), touched=int(c[6]), roots=int(c[7]))

    def map_tree_slide(self, position_last, sliding_thresh, half_map_size):
        """VoxelMapManager::mapSliding on the device tree.  Returns (root voxels removed or -1 below the threshold, dict of what the free stacks hold)."""
        pos = _f64(position_last).reshape(3)
        removed = C.c_int32(0)
        fc = np.zeros(3, np.int32)
        self._chk(self.lib.livo2_map_tree_slide(self.h, abi.as_ptr(pos, C.c_double), float(sliding_thresh), int(half_map_size), C.byref(removed), abi.as_ptr(fc, C.c_int32)))
        return int(removed.value), dict(nodes=int(fc[0]), planes=int(fc[1]), slabs=int(fc[2]))

    def map_tree_last_kernel_us(self):
        return float(self.lib.livo2_map_tree_last_kernel_us(self.h))

    def map_tree_export(self):
        """The device tree as flat-map arrays (dict with the FlatMap field names + node_temp)."""
        st = self.map_tree_stats()
        R, N, P = st["roots"], max(st["nodes"], 1), max(st["planes"], 1)
        o = dict(root_key=np.zeros((R, 3), np.int64), root_node=np.zeros(R, np.int32), root_center=np.zeros((R, 3)), root_quarter=np.zeros(R, np.float32),
                 node_plane=np.zeros(N, np.int32), node_child=np.zeros((N, 8), np.int32), plane_normal=np.zeros((P, 3)), plane_center=np.zeros((P, 3)),
                 plane_var=np.zeros((P, 36)), plane_d=np.zeros(P, np.float32), plane_radius=np.zeros(P, np.float32), node_temp=np.zeros(N, np.int32))
        self._chk(self.lib.livo2_map_tree_export(self.h, abi.as_ptr(o["root_key"], C.c_int64), abi.as_ptr(o["root_node"], C.c_int32), abi.as_ptr(o["root_center"], C.c_double),
                                                 abi.as_ptr(o["root_quarter"], C.c_float), abi.as_ptr(o["node_plane"], C.c_int32), abi.as_ptr(o["node_child"], C.c_int32),
                                                 abi.as_ptr(o["plane_normal"], C.c_double), abi.as_ptr(o["plane_center"], C.c_double), abi.as_ptr(o["plane_var"], C.c_double),
                                                 abi.as_ptr(o["plane_d"], C.c_float), abi.as_ptr(o["plane_radius"], C.c_float), abi.as_ptr(o["node_temp"], C.c_int32)))
        for k in ("node_plane", "node_child", "node_temp"):
            o[k] = o[k][: st["nodes"]]
        for k in ("plane_normal", "plane_center", "plane_var", "plane_d", "plane_radius"):
            o[k] = o[k][: st["planes"]]
        return o

    # ---- LiDAR -----------------------------------------------------------------------------------------------
    def set_scan(self, xyz, cfg):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.n = len(xyz)
        self._chk(self.lib.livo2_lidar_set_scan(self.h, abi.as_ptr(xyz, C.c_float), self.n, C.byref(cfg)))

    _POINT_FIELDS = {"match_plane": (np.int32, 1), "dis_to_plane": (np.float32, 1), "point_w": (np.float32, 3), "normal_plane": (np.int32, 1),
                     "var": (np.float64, 9), "body_cov": (np.float64, 9), "r_inv": (np.float64, 1), "h_row": (np.float64, 6)}

    def preprocess_scan(self, xyz, curvature, poses, rot_end, pos_end, leaf, cfg, want=True):
        """Raw scan -> undistortion -> voxel grid -> resident scan.  poses: [K][22] Pose6D rows.  Returns (n_down, feats_undistort, feats_down_body)."""
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        cur = np.ascontiguousarray(curvature, np.float32)
        P = np.ascontiguousarray(poses, np.float64).reshape(-1, 22)
        re, pe = _f64(rot_end), _f64(pos_end)
        n = len(x)
        und = np.zeros((n, 3), np.float32) if want else None
        down = np.zeros((max(n, 1), 3), np.float32) if want else None
        nd = C.c_int32(0)
        self._chk(self.lib.livo2_lidar_preprocess_scan(self.h, abi.as_ptr(x, C.c_float), abi.as_ptr(cur, C.c_float), n, P.ctypes.data_as(C.c_void_p), len(P),
                                                       abi.as_ptr(re, C.c_double), abi.as_ptr(pe, C.c_double), float(leaf), C.byref(cfg), C.byref(nd),
                                                       abi.as_ptr(und, C.c_float) if want else None, abi.as_ptr(down, C.c_float) if want else None))
        self.n = int(nd.value)
        return self.n, und, (down[: self.n] if want else None)

    def lio_frame(self, state_in, steps, imu_cfg, first_pose, xyz, curvature, leaf, cfg, want_poses=True):
        """One LiDAR-inertial frame in one call: IMU forward propagation -> undistortion + voxel grid -> StateEstimation(state_propagat).
        steps: [n][8]; first_pose: [22] (IMUpose[0]).  Returns (LidarResult, n_down, state_propagat, poses [n][22] or None)."""
        S = np.ascontiguousarray(steps, np.float64).reshape(-1, 8)
        fp = np.ascontiguousarray(first_pose, np.float64).reshape(22)
        x = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        cur = np.ascontiguousarray(curvature, np.float32)
        prop, res, nd = State(), LidarResult(), C.c_int32(0)
        poses = np.zeros((max(len(S), 1), 22)) if want_poses else None
        self._chk(self.lib.livo2_lio_frame(self.h, C.byref(state_in), S.ctypes.data_as(C.c_void_p), len(S), C.byref(imu_cfg), fp.ctypes.data_as(C.c_void_p),
                                           abi.as_ptr(x, C.c_float), abi.as_ptr(cur, C.c_float), len(x), float(leaf), C.byref(cfg), C.byref(prop),
                                           poses.ctypes.data_as(C.c_void_p) if want_poses else None, C.byref(nd), C.byref(res)))
        self.n = int(nd.value)
        return res, self.n, prop, (poses[: len(S)] if want_poses else None)

    def preprocess_last_kernel_us(self):
        return float(self.lib.livo2_lidar_preprocess_last_kernel_us(self.h))

    def _points(self, want):
        if not want:
            return None, {}
        pts, out = LidarPoints(), {}
        for name in want:
            dt, w = self._POINT_FIELDS[name]
            arr = np.zeros((self.n, w) if w > 1 else (self.n,), dt)
            out[name] = arr
            ct = {np.int32: C.c_int32, np.float32: C.c_float, np.float64: C.c_double}[dt]
            setattr(pts, name, abi.as_ptr(arr, ct))
        return pts, out

    def lidar_iterate(self, cur, prop, cfg, want=()):
        sums = LidarSums()
        pts, out = self._points(want)
        self._chk(self.lib.livo2_lidar_iterate(self.h, C.byref(cur), C.byref(prop), C.byref(cfg), C.byref(sums), C.byref(pts) if pts else None))
        return sums, out

    def lidar_update(self, state_in, prop, cfg, want=()):
        res = LidarResult()
        pts, out = self._points(want)
        self._chk(self.lib.livo2_lidar_update(self.h, C.byref(state_in), C.byref(prop), C.byref(cfg), C.byref(res), C.byref(pts) if pts else None))
        return res, out

    def lidar_update_async(self, state_in, prop, cfg):
        self._chk(self.lib.livo2_lidar_update_async(self.h, C.byref(state_in), C.byref(prop), C.byref(cfg), None))

    def lidar_update_fetch(self):
        res = LidarResult()
        self._chk(self.lib.livo2_lidar_update_fetch(self.h, C.byref(res), None))
        return res

    def lidar_iterations_async(self, state_in, prop, cfg, iters):
        self._chk(self.lib.livo2_lidar_iterations_async(self.h, C.byref(state_in), C.byref(prop), C.byref(cfg), int(iters)))

    # ---- IMU forward propagation ----------------------------------------------------------------------------------
    def imu_propagate(self, state_in, steps, cfg):
        """steps: [n][8] = gyr3, acc3, dt, offs_t (averaged raw measurements); cfg: abi.ImuCfg.  Returns (state_out, poses [n][22])."""
        S = np.ascontiguousarray(steps, np.float64).reshape(-1, 8)
        out = State()
        poses = np.zeros((max(len(S), 1), 22))
        self._chk(self.lib.livo2_imu_propagate(self.h, C.byref(state_in), S.ctypes.data_as(C.c_void_p), len(S), C.byref(cfg), C.byref(out), poses.ctypes.data_as(C.c_void_p)))
        return out, poses[: len(S)]

    def imu_last_kernel_us(self):
        return float(self.lib.livo2_imu_propagate_last_kernel_us(self.h))

    # ---- map maintenance ---------------------------------------------------------------------------------------
    def plane_fit_batch(self, point_w, var, offsets, planer_threshold, plane_idx=None):
        """VoxelOctoTree::init_plane for every group [offsets[g], offsets[g+1]); returns a ctypes array of PlaneFit."""
        pw = np.ascontiguousarray(point_w, np.float64).reshape(-1, 3)
        v = np.ascontiguousarray(var, np.float64).reshape(-1, 9)
        off = np.ascontiguousarray(offsets, np.int32)
        G = len(off) - 1
        out = (PlaneFit * max(G, 1))()
        idx = None if plane_idx is None else np.ascontiguousarray(plane_idx, np.int32)
        self._chk(self.lib.livo2_plane_fit_batch(self.h, abi.as_ptr(pw, C.c_double), abi.as_ptr(v, C.c_double), abi.as_ptr(off, C.c_int32), G,
                                                 float(planer_threshold), abi.as_ptr(idx, C.c_int32) if idx is not None else None, out))
        return out

    def plane_fit_last_kernel_us(self):
        return float(self.lib.livo2_plane_fit_last_kernel_us(self.h))

    # ---- batch of frames against the resident map --------------------------------------------------------------
    def batch_set_scans(self, scans, cfg):
        """scans: list of [n_f][3] float32 arrays (sensor frame), one per frame."""
        counts = np.array([len(x) for x in scans], np.int32)
        xyz = np.ascontiguousarray(np.concatenate([np.asarray(x, np.float32).reshape(-1, 3) for x in scans]), np.float32) if counts.sum() else np.zeros((0, 3), np.float32)
        self._chk(self.lib.livo2_lidar_batch_set_scans(self.h, len(scans), abi.as_ptr(xyz, C.c_float), abi.as_ptr(counts, C.c_int32), C.byref(cfg)))
        self.batch_n = len(scans)

    @staticmethod
    def _state_array(states):
        arr = (State * len(states))()
        for k, st in enumerate(states):
            C.memmove(C.byref(arr, k * C.sizeof(State)), C.byref(st), C.sizeof(State))
        return arr

    def batch_update(self, states_in, props, cfg):
        res = (LidarResult * self.batch_n)()
        self._chk(self.lib.livo2_lidar_batch_update(self.h, self.batch_n, self._state_array(states_in), self._state_array(props), C.byref(cfg), res))
        return list(res)

    def batch_update_async(self, states_in, props, cfg):
        self._chk(self.lib.livo2_lidar_batch_update_async(self.h, self.batch_n, self._state_array(states_in), self._state_array(props), C.byref(cfg)))

    def batch_update_fetch(self):
        res = (LidarResult * self.batch_n)()
        self._chk(self.lib.livo2_lidar_batch_update_fetch(self.h, self.batch_n, res))
        return list(res)

    def batch_iterations_async(self, states_in, props, cfg, iters):
        self._chk(self.lib.livo2_lidar_batch_iterations_async(self.h, self.batch_n, self._state_array(states_in), self._state_array(props), C.byref(cfg), int(iters)))

    # ---- visual ----------------------------------------------------------------------------------------------
    def set_frame(self, img, pos, warp_patch, search_levels, inv_expo_list):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        pos = _f64(pos).reshape(-1, 3)
        M = len(pos)
        warp_patch = np.ascontiguousarray(warp_patch, np.float32).reshape(M, -1, 64) if M else np.zeros((0, 1, 64), np.float32)
        L = warp_patch.shape[1]
        search_levels = np.ascontiguousarray(search_levels, np.int32)
        inv_expo_list = _f64(inv_expo_list)
        self.M, self.L = M, L
        self._chk(self.lib.livo2_visual_set_frame(self.h, abi.as_ptr(img, C.c_uint8), w, h, w, abi.as_ptr(pos, C.c_double), abi.as_ptr(warp_patch, C.c_float),
                                                  abi.as_ptr(search_levels, C.c_int32), abi.as_ptr(inv_expo_list, C.c_double), M, L))

    def visual_map_upload(self, pos, keys=None, active=None):
        pos = _f64(pos).reshape(-1, 3)
        k = None if keys is None else np.ascontiguousarray(keys, np.int64)
        a = None if active is None else np.ascontiguousarray(active, np.uint8)
        self.n_vm = len(pos)
        self._chk(self.lib.livo2_visual_map_upload(self.h, len(pos), abi.as_ptr(pos, C.c_double), abi.as_ptr(k, C.c_int64) if k is not None else None,
                                                   abi.as_ptr(a, C.c_uint8) if a is not None else None))

    def raycast_fetch(self, capacity=32768):
        """visual_submap->add_from_voxel_map of the last selection that ran with raycast_en: [n][6] = center_, normal_"""
        add, n = np.zeros((capacity, 6)), C.c_int32(0)
        self._chk(self.lib.livo2_visual_raycast_fetch(self.h, abi.as_ptr(add, C.c_double), capacity, C.byref(n)))
        return add[: n.value]

    def visual_select(self, ss, raycast=False):
        """Selection half of retrieveFromVisualSparseMap over a scenario-like object (pg, R_cur, t_cur, cam, border, grid_*).  raycast: vio/raycast_en (the RayCasting
        module looks into the device-resident VoxelMap of this context, if there is one); adds out["add_from_voxel_map"]."""
        c = SelectCfg()
        c.raycast_en = 1 if raycast else 0
        c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = ss.cam["fx"], ss.cam["fy"], ss.cam["cx"], ss.cam["cy"]
        c.cam.distortion, c.cam.width, c.cam.height = 0, ss.cam["width"], ss.cam["height"]
        _cam_distortion(c.cam, ss.cam)
        c.R_cur[:] = np.asarray(ss.R_cur, float).ravel().tolist(); c.t_cur[:] = np.asarray(ss.t_cur, float).tolist()
        c.border, c.grid_size, c.grid_n_width, c.grid_n_height, c.patch_size_half = int(ss.border), int(ss.grid_size), int(ss.grid_n_width), int(ss.grid_n_height), 4
        length = int(ss.grid_n_width) * int(ss.grid_n_height)
        pg = _f64(ss.pg).reshape(-1, 3)
        out = dict(cell_point=np.zeros(length, np.int32), cell_dist=np.zeros(length, np.float32), discont=np.zeros(length, np.uint8), in_fov=np.zeros(max(self.n_vm, 1), np.uint8))
        self._chk(self.lib.livo2_visual_select(self.h, abi.as_ptr(pg, C.c_double), len(pg), C.byref(c), abi.as_ptr(out["cell_point"], C.c_int32),
                                               abi.as_ptr(out["cell_dist"], C.c_float), abi.as_ptr(out["discont"], C.c_uint8), abi.as_ptr(out["in_fov"], C.c_uint8)))
        out["in_fov"] = out["in_fov"][: self.n_vm]
        if raycast:
            out["add_from_voxel_map"] = self.raycast_fetch(length)
        return out

    def select_last_kernel_us(self):
        return float(self.lib.livo2_visual_select_last_kernel_us(self.h))

    def retrieve_warp(self, rs, want_patches=True):
        """Per-point tail of retrieveFromVisualSparseMap over a scenario-like object (img, ref_imgs, pos, normal, ref_*, R_cur, t_cur,
        inv_expo_cur, cam, cfg); leaves the survivors resident as the frame.  Returns a dict of per-candidate arrays + n_accepted."""
        n, L = len(rs.pos), int(rs.cfg["patch_pyrimid_level"])
        c = RetrieveCfg()
        c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy = rs.cam["fx"], rs.cam["fy"], rs.cam["cx"], rs.cam["cy"]
        c.cam.distortion, c.cam.width, c.cam.height = 0, rs.cam["width"], rs.cam["height"]
        _cam_distortion(c.cam, rs.cam)
        c.R_cur[:] = np.asarray(rs.R_cur, float).ravel().tolist(); c.t_cur[:] = np.asarray(rs.t_cur, float).tolist(); c.inv_expo_cur = float(rs.inv_expo_cur)
        c.patch_pyrimid_level, c.normal_en, c.ncc_en = L, int(rs.cfg["normal_en"]), int(rs.cfg["ncc_en"])
        c.ncc_thre, c.outlier_threshold = float(rs.cfg["ncc_thre"]), float(rs.cfg["outlier_threshold"])
        img = np.ascontiguousarray(rs.img, np.uint8)
        refs = np.ascontiguousarray(rs.ref_imgs, np.uint8)
        keep = dict(pos=_f64(rs.pos), normal=_f64(rs.normal), px=_f64(rs.ref_px), f=_f64(rs.ref_f), R=_f64(rs.ref_R), t=_f64(rs.ref_t), ie=_f64(rs.ref_inv_expo),
                    idx=np.ascontiguousarray(rs.ref_img_idx, np.int32), lvl=np.ascontiguousarray(rs.ref_level, np.int32))
        rid = getattr(rs, "ref_id", None)
        if rid is not None:
            keep["id"] = np.ascontiguousarray(rid, np.int32)
        cand = RetrieveCandidates(n, 0, abi.as_ptr(keep["pos"], C.c_double), abi.as_ptr(keep["normal"], C.c_double), abi.as_ptr(keep["idx"], C.c_int32),
                                  abi.as_ptr(keep["px"], C.c_double), abi.as_ptr(keep["f"], C.c_double), abi.as_ptr(keep["R"], C.c_double),
                                  abi.as_ptr(keep["t"], C.c_double), abi.as_ptr(keep["lvl"], C.c_int32), abi.as_ptr(keep["ie"], C.c_double),
                                  abi.as_ptr(keep["id"], C.c_int32) if rid is not None else None)
        res = dict(accepted=np.zeros(n, np.int32), search_level=np.zeros(n, np.int32), error=np.zeros(n, np.float32), ncc=np.zeros(n), A=np.zeros((n, 4)),
                   patch_wrap=np.zeros((n, L, 64), np.float32) if want_patches else None)
        out = RetrieveOut(abi.as_ptr(res["accepted"], C.c_int32), abi.as_ptr(res["search_level"], C.c_int32), abi.as_ptr(res["error"], C.c_float),
                          abi.as_ptr(res["ncc"], C.c_double), abi.as_ptr(res["A"], C.c_double),
                          abi.as_ptr(res["patch_wrap"], C.c_float) if want_patches else None)
        na = C.c_int32(0)
        self._chk(self.lib.livo2_visual_retrieve_warp(self.h, abi.as_ptr(img, C.c_uint8), img.shape[1], img.shape[0], img.shape[1], abi.as_ptr(refs, C.c_uint8),
                                                      int(refs.shape[0]), C.byref(cand), C.byref(c), C.byref(out), C.byref(na)))
        res["n_accepted"] = int(na.value)
        self.M, self.L = int(na.value), L
        return res

    def retrieve_last_kernel_us(self):
        return float(self.lib.livo2_visual_retrieve_last_kernel_us(self.h))

    def visual_obs_upload(self, cs):
        """Observation table of the visual map (after visual_map_upload, same point order) from a scenario-like object with obs_offset, obs_*,
        normal, normal_initialized, ref_patch, ref_imgs (scenarios.synth.RetrieveChainScenario)."""
        refs = np.ascontiguousarray(cs.ref_imgs, np.uint8)
        k = dict(off=np.ascontiguousarray(cs.obs_offset, np.int32), id=np.ascontiguousarray(cs.obs_id, np.int32), img=np.ascontiguousarray(cs.obs_img_idx, np.int32),
                 px=_f64(cs.obs_px), f=_f64(cs.obs_f), R=_f64(cs.obs_R), t=_f64(cs.obs_t), lvl=np.ascontiguousarray(cs.obs_level, np.int32), ie=_f64(cs.obs_inv_expo),
                 patch=np.ascontiguousarray(cs.obs_patch, np.float32), normal=_f64(cs.normal), ninit=np.ascontiguousarray(cs.normal_initialized, np.uint8),
                 rp=np.ascontiguousarray(cs.ref_patch, np.int32))
        o = VisualObs(len(k["id"]), int(refs.shape[0]), abi.as_ptr(k["off"], C.c_int32), abi.as_ptr(k["id"], C.c_int32), abi.as_ptr(k["img"], C.c_int32),
                      abi.as_ptr(k["px"], C.c_double), abi.as_ptr(k["f"], C.c_double), abi.as_ptr(k["R"], C.c_double), abi.as_ptr(k["t"], C.c_double),
                      abi.as_ptr(k["lvl"], C.c_int32), abi.as_ptr(k["ie"], C.c_double), abi.as_ptr(k["patch"], C.c_float), abi.as_ptr(k["normal"], C.c_double),
                      abi.as_ptr(k["ninit"], C.c_uint8), abi.as_ptr(k["rp"], C.c_int32), abi.as_ptr(refs, C.c_uint8), int(refs.shape[2]), int(refs.shape[1]),
                      int(refs.shape[2]), 0)
        self._chk(self.lib.livo2_visual_obs_upload(self.h, C.byref(o)))

    def visual_map_counts(self):
        c = np.zeros(4, np.int32)
        self._chk(self.lib.livo2_visual_map_counts(self.h, abi.as_ptr(c, C.c_int32)))
        return dict(points=int(c[0]), obs=int(c[1]), ref_imgs=int(c[2]), stride=int(c[3]))

    def visual_map_apply(self, new_pos=None, new_keys=None, new_active=None, obs=None, touched=None, img=None, img_slot=0):
        """livo2_visual_map_apply: one frame's changes of the resident visual map.  obs: dict(id, img_idx, px, f, R, t, level, inv_expo, patch) of the NEW
        observations; touched: dict(point [q], lists = list of q index lists (global observation indices, obs_ order), normal [q,3], normal_initialized [q],
        ref_patch [q], active [q] optional); img: one new reference image [H,W] u8 for slot img_slot."""
        keep = []

        def arr(a, dt, ct):
            a = np.ascontiguousarray(a, dt); keep.append(a)
            return abi.as_ptr(a, ct)
        d = abi.VisualMapDelta()
        d.img_slot = int(img_slot)
        if new_pos is not None and len(new_pos):
            d.n_new_points = len(new_pos)
            d.new_pos = arr(np.asarray(new_pos, np.float64).reshape(-1, 3), np.float64, C.c_double); d.new_voxel_key = arr(np.asarray(new_keys).reshape(-1, 3), np.int64, C.c_int64)
            if new_active is not None:
                d.new_active = arr(new_active, np.uint8, C.c_uint8)
        if obs is not None and len(obs["id"]):
            d.n_new_obs = len(obs["id"])
            d.obs_id = arr(obs["id"], np.int32, C.c_int32); d.obs_img_idx = arr(obs["img_idx"], np.int32, C.c_int32); d.obs_level = arr(obs["level"], np.int32, C.c_int32)
            d.obs_px = arr(obs["px"], np.float64, C.c_double); d.obs_f = arr(obs["f"], np.float64, C.c_double); d.obs_R = arr(obs["R"], np.float64, C.c_double)
            d.obs_t = arr(obs["t"], np.float64, C.c_double); d.obs_inv_expo = arr(obs["inv_expo"], np.float64, C.c_double); d.obs_patch = arr(obs["patch"], np.float32, C.c_float)
        if touched is not None and len(touched["point"]):
            q = len(touched["point"])
            d.n_touched = q
            off = np.zeros(q + 1, np.int32); off[1:] = np.cumsum([len(l) for l in touched["lists"]])
            flat = np.concatenate([np.asarray(l, np.int32) for l in touched["lists"]]) if off[-1] else np.zeros(1, np.int32)
            d.touched_point = arr(touched["point"], np.int32, C.c_int32); d.touched_offset = arr(off, np.int32, C.c_int32); d.touched_obs = arr(flat, np.int32, C.c_int32)
            d.touched_normal = arr(np.asarray(touched["normal"], np.float64).reshape(-1, 3), np.float64, C.c_double)
            d.touched_normal_initialized = arr(touched["normal_initialized"], np.uint8, C.c_uint8); d.touched_ref_patch = arr(touched["ref_patch"], np.int32, C.c_int32)
            if touched.get("active") is not None:
                d.touched_active = arr(touched["active"], np.uint8, C.c_uint8)
        if img is not None:
            d.img = arr(img, np.uint8, C.c_uint8)
        self._chk(self.lib.livo2_visual_map_apply(self.h, C.byref(d)))
        self.n_vm += int(d.n_new_points)

    def visual_retrieve_from_map(self, cs, want_patches=True, raycast=False):
        """The whole retrieveFromVisualSparseMap as one chain (selection -> reference-patch choice -> tail) over a RetrieveChainScenario-like object
        whose points / observations were uploaded with visual_map_upload + visual_obs_upload.  Returns the stage outputs (same names as
        oracle.orc.visual_retrieve) and leaves the survivors resident as the frame."""
        ss, L = cs.sel, int(cs.cfg["patch_pyrimid_level"])
        sc = SelectCfg()
        sc.cam.fx, sc.cam.fy, sc.cam.cx, sc.cam.cy = ss.cam["fx"], ss.cam["fy"], ss.cam["cx"], ss.cam["cy"]
        sc.cam.distortion, sc.cam.width, sc.cam.height = 0, ss.cam["width"], ss.cam["height"]
        _cam_distortion(sc.cam, ss.cam)
        sc.R_cur[:] = np.asarray(ss.R_cur, float).ravel().tolist(); sc.t_cur[:] = np.asarray(ss.t_cur, float).tolist()
        sc.border, sc.grid_size, sc.grid_n_width, sc.grid_n_height, sc.patch_size_half = int(ss.border), int(ss.grid_size), int(ss.grid_n_width), int(ss.grid_n_height), 4
        sc.raycast_en = 1 if raycast else 0
        c = RetrieveCfg()
        c.cam = sc.cam
        c.R_cur[:] = np.asarray(ss.R_cur, float).ravel().tolist(); c.t_cur[:] = np.asarray(ss.t_cur, float).tolist(); c.inv_expo_cur = float(cs.inv_expo_cur)
        c.patch_pyrimid_level, c.normal_en, c.ncc_en = L, int(cs.cfg["normal_en"]), int(cs.cfg["ncc_en"])
        c.ncc_thre, c.outlier_threshold = float(cs.cfg["ncc_thre"]), float(cs.cfg["outlier_threshold"])
        length = int(ss.grid_n_width) * int(ss.grid_n_height)
        img, pg = np.ascontiguousarray(cs.img, np.uint8), _f64(ss.pg).reshape(-1, 3)
        r = dict(cell_point=np.zeros(length, np.int32), cell_dist=np.zeros(length, np.float32), discont=np.zeros(length, np.uint8), cell_obs=np.zeros(length, np.int32),
                 ref_patch=np.zeros(max(self.n_vm, 1), np.int32), cand_cell=np.zeros(length, np.int32), sub_point=np.zeros(length, np.int32), sub_obs=np.zeros(length, np.int32))
        t = dict(accepted=np.zeros(length, np.int32), search_level=np.zeros(length, np.int32), error=np.zeros(length, np.float32), ncc=np.zeros(length), A=np.zeros((length, 4)),
                 patch_wrap=np.zeros((length, L, 64), np.float32) if want_patches else None)
        out = RetrieveChainOut(abi.as_ptr(r["cell_point"], C.c_int32), abi.as_ptr(r["cell_dist"], C.c_float), abi.as_ptr(r["discont"], C.c_uint8),
                               abi.as_ptr(r["cell_obs"], C.c_int32), abi.as_ptr(r["ref_patch"], C.c_int32), abi.as_ptr(r["cand_cell"], C.c_int32),
                               RetrieveOut(abi.as_ptr(t["accepted"], C.c_int32), abi.as_ptr(t["search_level"], C.c_int32), abi.as_ptr(t["error"], C.c_float),
                                           abi.as_ptr(t["ncc"], C.c_double), abi.as_ptr(t["A"], C.c_double), abi.as_ptr(t["patch_wrap"], C.c_float) if want_patches else None),
                               abi.as_ptr(r["sub_point"], C.c_int32), abi.as_ptr(r["sub_obs"], C.c_int32))
        nc, na = C.c_int32(0), C.c_int32(0)
        self._chk(self.lib.livo2_visual_retrieve_from_map(self.h, abi.as_ptr(img, C.c_uint8), img.shape[1], img.shape[0], img.shape[1], abi.as_ptr(pg, C.c_double), len(pg),
                                                          C.byref(sc), C.byref(c), C.byref(out), C.byref(nc), C.byref(na)))
        nc, na = int(nc.value), int(na.value)
        r["ref_patch"] = r["ref_patch"][: self.n_vm]
        r["cand_cell"], r["sub_point"], r["sub_obs"] = r["cand_cell"][:nc], r["sub_point"][:na], r["sub_obs"][:na]
        r["tail"] = {k: (v[:nc] if v is not None else None) for k, v in t.items()}
        r["n_candidates"], r["n_accepted"] = nc, na
        self.M, self.L = na, L
        if raycast:
            r["add_from_voxel_map"] = self.raycast_fetch(length)
        return r

    def retrieve_from_map_last_kernel_us(self):
        return float(self.lib.livo2_visual_retrieve_from_map_last_kernel_us(self.h))

    def set_reference(self, ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos):
        """inverse-compositional variant: reference patches of the points uploaded by set_frame"""
        ref_imgs = np.ascontiguousarray(ref_imgs, np.uint8)
        if ref_imgs.ndim == 2:
            ref_imgs = ref_imgs[None]
        idx = np.ascontiguousarray(ref_img_idx, np.int32)
        px, f, R, pos = _f64(ref_px).reshape(-1, 2), _f64(ref_f).reshape(-1, 3), _f64(ref_R).reshape(-1, 9), _f64(ref_pos).reshape(-1, 3)
        self._chk(self.lib.livo2_visual_set_reference(self.h, abi.as_ptr(ref_imgs, C.c_uint8), len(ref_imgs), abi.as_ptr(idx, C.c_int32), abi.as_ptr(px, C.c_double),
                                                      abi.as_ptr(f, C.c_double), abi.as_ptr(R, C.c_double), abi.as_ptr(pos, C.c_double)))

    def visual_iterate(self, level, cur, cfg, rows=False):
        sums = VisualSums()
        errors = np.zeros(self.M, np.float32)
        z = np.zeros(self.M * 64, np.float64) if rows else None
        H = np.zeros((self.M * 64, 7), np.float64) if rows else None
        self._chk(self.lib.livo2_visual_iterate(self.h, int(level), C.byref(cur), C.byref(cfg), C.byref(sums), abi.as_ptr(errors, C.c_float),
                                                abi.as_ptr(z, C.c_double) if rows else None, abi.as_ptr(H, C.c_double) if rows else None))
        return sums, errors, z, H

    def visual_update(self, state_in, prop, cfg):
        res = VisualResult()
        errors = np.zeros(self.M, np.float32)
        self._chk(self.lib.livo2_visual_update(self.h, C.byref(state_in), C.byref(prop), C.byref(cfg), C.byref(res), abi.as_ptr(errors, C.c_float)))
        return res, errors

    # ---- one LIO + VIO frame per call ------------------------------------------------------------------------------
    def _frame_in(self, xyz, prior, cfg, vs_img, pos, warp_patch, search_levels, inv_expo_list, vcfg, reference=None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        img = np.ascontiguousarray(vs_img, np.uint8)
        pos = _f64(pos).reshape(-1, 3)
        warp = np.ascontiguousarray(warp_patch, np.float32)
        sl, ie = np.ascontiguousarray(search_levels, np.int32), _f64(inv_expo_list)
        M = len(pos)
        L = int(warp.shape[1]) if warp.ndim == 3 else int(vcfg.patch_pyrimid_level)
        f = abi.FrameIn(abi.as_ptr(xyz, C.c_float), len(xyz), M, L, img.shape[1], img.shape[0], img.shape[1], C.addressof(prior), C.addressof(cfg), C.addressof(vcfg),
                        abi.as_ptr(img, C.c_uint8), abi.as_ptr(pos, C.c_double), abi.as_ptr(warp, C.c_float), abi.as_ptr(sl, C.c_int32), abi.as_ptr(ie, C.c_double), None)
        keep_ref = None
        if reference is not None:          # (ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos) as set_reference takes them
            ri = np.ascontiguousarray(reference[0], np.uint8)
            ri = ri[None] if ri.ndim == 2 else ri
            idx = np.ascontiguousarray(reference[1], np.int32)
            px, ff, RR, pp = _f64(reference[2]).reshape(-1, 2), _f64(reference[3]).reshape(-1, 3), _f64(reference[4]).reshape(-1, 9), _f64(reference[5]).reshape(-1, 3)
            ref = abi.VisualReference(abi.as_ptr(ri, C.c_uint8), len(ri), 0, abi.as_ptr(idx, C.c_int32), abi.as_ptr(px, C.c_double), abi.as_ptr(ff, C.c_double),
                                      abi.as_ptr(RR, C.c_double), abi.as_ptr(pp, C.c_double))
            f.reference = C.addressof(ref)
            keep_ref = (ref, ri, idx, px, ff, RR, pp)
        return f, (xyz, img, pos, warp, sl, ie, prior, cfg, vcfg, keep_ref), M, L

    def frame_update_async(self, xyz, prior, cfg, img, pos, warp_patch, search_levels, inv_expo_list, vcfg, reference=None):
        """livo2_frame_update_async: scan + StateEstimation from `prior` + image / sub-map + computeJacobianAndUpdateEKF from the LiDAR posterior, enqueued in one
        call (up to two frames in flight; the arrays are staged inside the call)."""
        f, keep, M, L = self._frame_in(xyz, prior, cfg, img, pos, warp_patch, search_levels, inv_expo_list, vcfg, reference)
        self._chk(self.lib.livo2_frame_update_async(self.h, C.byref(f)))
        self.n, self.M, self.L = len(keep[0]), M, L

    def frame_enqueue(self, fin):
        """livo2_frame_update_async on a prepared livo2_frame_in (_frame_in): what a timed loop calls"""
        self._chk(self.lib.livo2_frame_update_async(self.h, C.byref(fin)))

    def frame_update_fetch(self, into=None):
        lres, vres = into if into is not None else (LidarResult(), VisualResult())
        self._chk(self.lib.livo2_frame_update_fetch(self.h, C.byref(lres), C.byref(vres)))
        return lres, vres

    def new_frame_results(self):
        return LidarResult(), VisualResult()

    def frame_update(self, *a, **kw):
        self.frame_update_async(*a, **kw)
        return self.frame_update_fetch()

    def visual_update_async(self, state_in, prop, cfg):
        self._chk(self.lib.livo2_visual_update_async(self.h, C.byref(state_in), C.byref(prop), C.byref(cfg)))

    def visual_update_fetch(self):
        res = VisualResult()
        self._chk(self.lib.livo2_visual_update_fetch(self.h, C.byref(res), None))
        return res

    # ---- batch of frames, visual -----------------------------------------------------------------------------------
    def visual_batch_set_frames(self, frames):
        """frames: list of (img [h][w] u8, pos [M][3], warp_patch [M][L][64], search_levels [M], inv_expo_list [M]); all images of one size, one L."""
        imgs = np.ascontiguousarray(np.stack([np.asarray(f[0], np.uint8) for f in frames]))
        h, w = imgs.shape[1:]
        counts = np.array([len(f[1]) for f in frames], np.int32)
        L = max([np.asarray(f[2]).reshape(len(f[1]), -1, 64).shape[1] for f in frames if len(f[1])] or [1])
        pos = _f64(np.concatenate([np.asarray(f[1], np.float64).reshape(-1, 3) for f in frames]))
        warp = np.ascontiguousarray(np.concatenate([np.asarray(f[2], np.float32).reshape(len(f[1]), L, 64) for f in frames]), np.float32)
        sl = np.ascontiguousarray(np.concatenate([np.asarray(f[3], np.int32).reshape(-1) for f in frames]), np.int32)
        ie = _f64(np.concatenate([np.asarray(f[4], np.float64).reshape(-1) for f in frames]))
        self._chk(self.lib.livo2_visual_batch_set_frames(self.h, len(frames), abi.as_ptr(imgs, C.c_uint8), w, h, w, abi.as_ptr(pos, C.c_double), abi.as_ptr(warp, C.c_float),
                                                         abi.as_ptr(sl, C.c_int32), abi.as_ptr(ie, C.c_double), abi.as_ptr(counts, C.c_int32), L))
        self.vbatch_n = len(frames)

    def visual_batch_set_references(self, refs):
        """refs: per frame (ref_imgs [n_ref][h][w] u8, ref_img_idx [M], ref_px [M][2], ref_f [M][3], ref_R [M][3][3], ref_pos [M][3]) — the arguments of set_reference"""
        n_ref = np.array([len(r[0]) for r in refs], np.int32)
        imgs = np.ascontiguousarray(np.concatenate([np.asarray(r[0], np.uint8) for r in refs]))
        idx = np.ascontiguousarray(np.concatenate([np.asarray(r[1], np.int32).reshape(-1) for r in refs]), np.int32)
        px = _f64(np.concatenate([np.asarray(r[2], np.float64).reshape(-1, 2) for r in refs]))
        f = _f64(np.concatenate([np.asarray(r[3], np.float64).reshape(-1, 3) for r in refs]))
        R = _f64(np.concatenate([np.asarray(r[4], np.float64).reshape(-1, 9) for r in refs]))
        pos = _f64(np.concatenate([np.asarray(r[5], np.float64).reshape(-1, 3) for r in refs]))
        self._chk(self.lib.livo2_visual_batch_set_references(self.h, len(refs), abi.as_ptr(imgs, C.c_uint8), abi.as_ptr(n_ref, C.c_int32), abi.as_ptr(idx, C.c_int32),
                                                             abi.as_ptr(px, C.c_double), abi.as_ptr(f, C.c_double), abi.as_ptr(R, C.c_double), abi.as_ptr(pos, C.c_double)))

    def visual_batch_update(self, states_in, props, cfg):
        res = (VisualResult * self.vbatch_n)()
        self._chk(self.lib.livo2_visual_batch_update(self.h, self.vbatch_n, self._state_array(states_in), self._state_array(props), C.byref(cfg), res))
        return list(res)

    def visual_batch_update_async(self, states_in, props, cfg):
        self._chk(self.lib.livo2_visual_batch_update_async(self.h, self.vbatch_n, self._state_array(states_in), self._state_array(props), C.byref(cfg)))

    def visual_batch_update_fetch(self):
        res = (VisualResult * self.vbatch_n)()
        self._chk(self.lib.livo2_visual_batch_update_fetch(self.h, self.vbatch_n, res))
        return list(res)

    def visual_batch_iterations_async(self, level, states_in, props, cfg, iters):
        self._chk(self.lib.livo2_visual_batch_iterations_async(self.h, self.vbatch_n, int(level), self._state_array(states_in), self._state_array(props), C.byref(cfg), int(iters)))

    def visual_iterations_async(self, level, state_in, prop, cfg, iters):
        self._chk(self.lib.livo2_visual_iterations_async(self.h, int(level), C.byref(state_in), C.byref(prop), C.byref(cfg), int(iters)))

    def debug_float_chain(self, errors, threads, lanes_per_chain=32):
        """Per-thread partial sums of the frame error (vio.cpp:1554, 1634): (by float_chain_wave, by the one-lane chain)."""
        e = np.ascontiguousarray(errors, np.float32)
        a, b = np.zeros(threads, np.float32), np.zeros(threads, np.float32)
        self._chk(self.lib.livo2_debug_float_chain(self.h, abi.as_ptr(e, C.c_float), len(e), int(threads), int(lanes_per_chain), abi.as_ptr(a, C.c_float),
                                                   abi.as_ptr(b, C.c_float)))
        return a, b

    # ---- solve ---------------------------------------------------------------------------------------------
    def esikf_solve(self, HtH, Htz, k, meas_cov_scale, sign, cur, prop):
        HtH, Htz = _f64(HtH).reshape(k, k), _f64(Htz).reshape(k)
        out, sol, G = State(), np.zeros(19), np.zeros((19, 19))
        self._chk(self.lib.livo2_esikf_solve(self.h, abi.as_ptr(HtH, C.c_double), abi.as_ptr(Htz, C.c_double), k, float(meas_cov_scale), int(sign),
                                             C.byref(cur), C.byref(prop), C.byref(out), abi.as_ptr(sol, C.c_double), abi.as_ptr(G, C.c_double)))
        return out, sol, G
