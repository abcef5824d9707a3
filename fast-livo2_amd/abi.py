"""ctypes binding of the C ABI declared in include/livo2_hip.h (liblivo2_hip.so).

This module is plumbing only: it mirrors the C structs, loads the in-tree shared library and fails loudly when the
HIP library is missing — there is no CPU fallback of any kind on the product path.
"""
import ctypes as C
import os

import numpy as np

DIM_STATE = 19
MAX_ITERS = 16
MAX_LEVELS = 8
MAX_LAYER = 4

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NO_MAP, ERR_NO_SCAN, ERR_NO_FRAME, ERR_RANGE = -1, -2, -3, -4, -5, -6, -7

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblivo2_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "livo2_hip.h")


class State(C.Structure):
    """livo2_state == StatesGroup (reference include/common_lib.h:126-223)."""
    _fields_ = [("rot", C.c_double * 9), ("pos", C.c_double * 3), ("inv_expo", C.c_double), ("vel", C.c_double * 3),
                ("bg", C.c_double * 3), ("ba", C.c_double * 3), ("grav", C.c_double * 3), ("cov", C.c_double * 361)]

    @staticmethod
    def default():
        """StatesGroup() (common_lib.h:128-140)."""
        s = State()
        s.rot[:] = np.eye(3).ravel().tolist()
        s.inv_expo = 1.0
        cov = np.eye(19) * 0.01
        cov[6, 6] = 0.00001
        cov[10:19, 10:19] = np.eye(9) * 0.00001
        s.cov[:] = cov.ravel().tolist()
        return s

    @staticmethod
    def from_pose(R, t, P, inv_expo=1.0):
        """state at pose (R, t) with covariance P; velocity, biases and gravity zero (what the update does not read)."""
        s = State()
        s.rot[:] = np.asarray(R, float).ravel().tolist()
        s.pos[:] = np.asarray(t, float).tolist()
        s.inv_expo = float(inv_expo)
        s.cov[:] = np.asarray(P, float).ravel().tolist()
        return s

    def copy(self):
        o = State()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(State))
        return o

    # numpy views -----------------------------------------------------------------------------------------------
    @property
    def R(self):
        return np.ctypeslib.as_array(self.rot).reshape(3, 3)

    @property
    def t(self):
        return np.ctypeslib.as_array(self.pos)

    @property
    def P(self):
        return np.ctypeslib.as_array(self.cov).reshape(19, 19)

    def vector(self):
        """all 25 non-covariance scalars: rot(9) pos(3) inv_expo vel bg ba grav"""
        return np.concatenate([np.array(self.rot), np.array(self.pos), [self.inv_expo], np.array(self.vel), np.array(self.bg),
                               np.array(self.ba), np.array(self.grav)])


class MapView(C.Structure):
    _fields_ = [("n_roots", C.c_int32), ("n_nodes", C.c_int32), ("n_planes", C.c_int32),
                ("root_key", C.POINTER(C.c_int64)), ("root_node", C.POINTER(C.c_int32)), ("root_center", C.POINTER(C.c_double)),
                ("root_quarter", C.POINTER(C.c_float)), ("node_plane", C.POINTER(C.c_int32)), ("node_child", C.POINTER(C.c_int32)),
                ("plane_normal", C.POINTER(C.c_double)), ("plane_center", C.POINTER(C.c_double)), ("plane_var", C.POINTER(C.c_double)),
                ("plane_d", C.POINTER(C.c_float)), ("plane_radius", C.POINTER(C.c_float))]


class LidarCfg(C.Structure):
    _fields_ = [("max_iterations", C.c_int32), ("max_layer", C.c_int32), ("sigma_num", C.c_double), ("dept_err", C.c_double),
                ("beam_err", C.c_double), ("voxel_size", C.c_double), ("deg2rad", C.c_double), ("extR", C.c_double * 9), ("extT", C.c_double * 3)]


class LidarSums(C.Structure):
    _fields_ = [("HtH", C.c_double * 36), ("Htz", C.c_double * 6), ("total_residual", C.c_double), ("n_eff", C.c_int32), ("pad", C.c_int32)]


class LidarPoints(C.Structure):
    _fields_ = [("match_plane", C.POINTER(C.c_int32)), ("dis_to_plane", C.POINTER(C.c_float)), ("point_w", C.POINTER(C.c_float)),
                ("normal_plane", C.POINTER(C.c_int32)), ("var", C.POINTER(C.c_double)), ("body_cov", C.POINTER(C.c_double)),
                ("r_inv", C.POINTER(C.c_double)), ("h_row", C.POINTER(C.c_double)), ("pinned", C.c_int64)]


class LidarResult(C.Structure):
    _fields_ = [("state", State), ("n_iters", C.c_int32), ("converged", C.c_int32), ("iter_sums", LidarSums * MAX_ITERS),
                ("iter_solution", (C.c_double * DIM_STATE) * MAX_ITERS), ("position_last", C.c_double * 3)]


class PlaneFit(C.Structure):
    """livo2_plane_fit == the VoxelPlane members init_plane writes (reference include/voxel_map.h:69-94)."""
    _fields_ = [("center", C.c_double * 3), ("normal", C.c_double * 3), ("y_normal", C.c_double * 3), ("x_normal", C.c_double * 3),
                ("covariance", C.c_double * 9), ("plane_var", C.c_double * 36), ("radius", C.c_float), ("min_eigen_value", C.c_float),
                ("mid_eigen_value", C.c_float), ("max_eigen_value", C.c_float), ("d", C.c_float), ("points_size", C.c_int32),
                ("is_plane", C.c_int32), ("pad", C.c_int32)]


class Cam(C.Structure):
    _fields_ = [("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("d", C.c_double * 5),
                ("distortion", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("pad", C.c_int32)]


class RetrieveCfg(C.Structure):
    _fields_ = [("cam", Cam), ("R_cur", C.c_double * 9), ("t_cur", C.c_double * 3), ("inv_expo_cur", C.c_double), ("patch_pyrimid_level", C.c_int32),
                ("normal_en", C.c_int32), ("ncc_en", C.c_int32), ("pad", C.c_int32), ("ncc_thre", C.c_double), ("outlier_threshold", C.c_double)]


class RetrieveCandidates(C.Structure):
    _fields_ = [("n", C.c_int32), ("pad", C.c_int32), ("pos", C.POINTER(C.c_double)), ("normal", C.POINTER(C.c_double)), ("ref_img_idx", C.POINTER(C.c_int32)),
                ("ref_px", C.POINTER(C.c_double)), ("ref_f", C.POINTER(C.c_double)), ("ref_R", C.POINTER(C.c_double)), ("ref_t", C.POINTER(C.c_double)), ("ref_level", C.POINTER(C.c_int32)),
                ("ref_inv_expo", C.POINTER(C.c_double)), ("ref_id", C.POINTER(C.c_int32))]


class RetrieveOut(C.Structure):
    _fields_ = [("accepted", C.POINTER(C.c_int32)), ("search_level", C.POINTER(C.c_int32)), ("error", C.POINTER(C.c_float)), ("ncc", C.POINTER(C.c_double)),
                ("A_cur_ref", C.POINTER(C.c_double)), ("patch_wrap", C.POINTER(C.c_float))]


class VisualObs(C.Structure):
    """livo2_visual_obs: CSR mirror of VisualPoint::obs_ (include/feature.h, include/visual_point.h of the reference)"""
    _fields_ = [("n_obs", C.c_int32), ("n_ref", C.c_int32), ("point_offset", C.POINTER(C.c_int32)), ("id", C.POINTER(C.c_int32)), ("img_idx", C.POINTER(C.c_int32)),
                ("px", C.POINTER(C.c_double)), ("f", C.POINTER(C.c_double)), ("R", C.POINTER(C.c_double)), ("t", C.POINTER(C.c_double)), ("level", C.POINTER(C.c_int32)),
                ("inv_expo", C.POINTER(C.c_double)), ("patch", C.POINTER(C.c_float)), ("normal", C.POINTER(C.c_double)), ("normal_initialized", C.POINTER(C.c_uint8)),
                ("ref_patch", C.POINTER(C.c_int32)), ("ref_imgs", C.POINTER(C.c_uint8)), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32)]


class VisualMapDelta(C.Structure):
    """livo2_visual_map_delta: one frame's changes of the visual map (generateVisualMapPoints / updateVisualMapPoints / updateReferencePatch of the reference)"""
    _fields_ = [("n_new_points", C.c_int32), ("n_new_obs", C.c_int32), ("n_touched", C.c_int32), ("img_slot", C.c_int32),
                ("new_pos", C.POINTER(C.c_double)), ("new_voxel_key", C.POINTER(C.c_int64)), ("new_active", C.POINTER(C.c_uint8)),
                ("obs_id", C.POINTER(C.c_int32)), ("obs_img_idx", C.POINTER(C.c_int32)), ("obs_px", C.POINTER(C.c_double)), ("obs_f", C.POINTER(C.c_double)),
                ("obs_R", C.POINTER(C.c_double)), ("obs_t", C.POINTER(C.c_double)), ("obs_level", C.POINTER(C.c_int32)), ("obs_inv_expo", C.POINTER(C.c_double)),
                ("obs_patch", C.POINTER(C.c_float)), ("touched_point", C.POINTER(C.c_int32)), ("touched_offset", C.POINTER(C.c_int32)), ("touched_obs", C.POINTER(C.c_int32)),
                ("touched_normal", C.POINTER(C.c_double)), ("touched_normal_initialized", C.POINTER(C.c_uint8)), ("touched_active", C.POINTER(C.c_uint8)),
                ("touched_ref_patch", C.POINTER(C.c_int32)), ("img", C.POINTER(C.c_uint8))]


class VisualReference(C.Structure):
    """livo2_visual_reference: reference patches of a frame's sub-map (inverse-compositional update)"""
    _fields_ = [("ref_imgs", C.POINTER(C.c_uint8)), ("n_ref", C.c_int32), ("pad", C.c_int32), ("ref_img_idx", C.POINTER(C.c_int32)), ("ref_px", C.POINTER(C.c_double)),
                ("ref_f", C.POINTER(C.c_double)), ("ref_R", C.POINTER(C.c_double)), ("ref_pos", C.POINTER(C.c_double))]


class FrameIn(C.Structure):
    """livo2_frame_in: one LIO + VIO frame (scan, prior, image + visual sub-map, both configurations)"""
    _fields_ = [("xyz", C.POINTER(C.c_float)), ("n_points", C.c_int32), ("M", C.c_int32), ("L", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32),
                ("prior", C.c_void_p), ("lidar_cfg", C.c_void_p), ("visual_cfg", C.c_void_p), ("img", C.POINTER(C.c_uint8)), ("pos", C.POINTER(C.c_double)),
                ("warp_patch", C.POINTER(C.c_float)), ("search_levels", C.POINTER(C.c_int32)), ("inv_expo_list", C.POINTER(C.c_double)), ("reference", C.c_void_p)]


class RetrieveChainOut(C.Structure):
    _fields_ = [("cell_point", C.POINTER(C.c_int32)), ("cell_dist", C.POINTER(C.c_float)), ("cell_discontinuous", C.POINTER(C.c_uint8)), ("cell_obs", C.POINTER(C.c_int32)),
                ("ref_patch", C.POINTER(C.c_int32)), ("cand_cell", C.POINTER(C.c_int32)), ("tail", RetrieveOut), ("sub_point", C.POINTER(C.c_int32)),
                ("sub_obs", C.POINTER(C.c_int32))]


class ImuCfg(C.Structure):
    _fields_ = [("cov_gyr", C.c_double * 3), ("cov_acc", C.c_double * 3), ("cov_bias_gyr", C.c_double * 3), ("cov_bias_acc", C.c_double * 3), ("cov_inv_expo", C.c_double),
                ("G_m_s2", C.c_double), ("mean_acc_norm", C.c_double), ("ba_bg_est_en", C.c_int32), ("gravity_est_en", C.c_int32), ("exposure_estimate_en", C.c_int32),
                ("first_call", C.c_int32)]


class SelectCfg(C.Structure):
    _fields_ = [("cam", Cam), ("R_cur", C.c_double * 9), ("t_cur", C.c_double * 3), ("border", C.c_int32), ("grid_size", C.c_int32), ("grid_n_width", C.c_int32),
                ("grid_n_height", C.c_int32), ("patch_size_half", C.c_int32), ("raycast_en", C.c_int32)]


class VisualCfg(C.Structure):
    _fields_ = [("cam", Cam), ("Rcl", C.c_double * 9), ("Pcl", C.c_double * 3), ("extR", C.c_double * 9), ("extT", C.c_double * 3),
                ("img_point_cov", C.c_double), ("patch_pyrimid_level", C.c_int32), ("max_iterations", C.c_int32),
                ("exposure_estimate_en", C.c_int32), ("inverse_composition_en", C.c_int32), ("mp_proc_num", C.c_int32), ("pad", C.c_int32)]


class MapTreeCfg(C.Structure):
    _fields_ = [("voxel_size", C.c_double), ("planer_threshold", C.c_double), ("max_layer", C.c_int32), ("max_points_num", C.c_int32),
                ("layer_init_num", C.c_int32 * 5), ("max_roots", C.c_int32), ("max_nodes", C.c_int32), ("max_planes", C.c_int32), ("max_points", C.c_int32),
                ("max_cand", C.c_int32)]


class VisualSums(C.Structure):
    _fields_ = [("HtH", C.c_double * 49), ("Htz", C.c_double * 7), ("err_sum", C.c_double), ("error", C.c_float), ("n_meas", C.c_int32)]


class VisualStep(C.Structure):
    _fields_ = [("level", C.c_int32), ("iteration", C.c_int32), ("accepted", C.c_int32), ("n_meas", C.c_int32), ("error", C.c_float),
                ("pad", C.c_int32), ("HtH", C.c_double * 49), ("Htz", C.c_double * 7), ("solution", C.c_double * DIM_STATE)]


class VisualResult(C.Structure):
    _fields_ = [("state", State), ("G", C.c_double * 361), ("Rcw", C.c_double * 9), ("Pcw", C.c_double * 3), ("n_steps", C.c_int32),
                ("pad", C.c_int32), ("steps", VisualStep * (MAX_LEVELS * MAX_ITERS))]


# name -> (restype, argtypes); every symbol include/livo2_hip.h declares
_P = C.POINTER
_CTX = C.c_void_p
SIGNATURES = {
    "livo2_ctx_create": (C.c_int, [C.c_int, _P(_CTX)]),
    "livo2_ctx_create_on_stream": (C.c_int, [C.c_int, C.c_void_p, _P(_CTX)]),
    "livo2_ctx_destroy": (None, [_CTX]),
    "livo2_last_error": (C.c_char_p, [_CTX]),
    "livo2_ctx_stream": (C.c_void_p, [_CTX]),
    "livo2_ctx_synchronize": (C.c_int, [_CTX]),
    "livo2_version": (C.c_char_p, []),
    "livo2_abi_sizeof": (C.c_int32, [C.c_char_p]),
    "livo2_ctx_kernel_timing": (C.c_int, [_CTX, C.c_int]),
    "livo2_ctx_set_option": (C.c_int, [_CTX, C.c_char_p, C.c_int32]),
    "livo2_ctx_get_counter": (C.c_int, [_CTX, C.c_char_p, _P(C.c_int64)]),
    "livo2_visual_raycast_fetch": (C.c_int, [_CTX, _P(C.c_double), C.c_int32, _P(C.c_int32)]),
    "livo2_host_alloc_pinned": (C.c_int, [C.c_size_t, _P(C.c_void_p)]),
    "livo2_host_free_pinned": (None, [C.c_void_p]),
    "livo2_debug_redzone_check": (C.c_int, [_CTX, _P(C.c_int32), _P(C.c_int64)]),
    "livo2_debug_redzone_poke": (C.c_int, [_CTX, C.c_int64]),
    "livo2_debug_float_chain": (C.c_int, [_CTX, _P(C.c_float), C.c_int32, C.c_int32, C.c_int32, _P(C.c_float), _P(C.c_float)]),
    "livo2_ctx_kernel_timing_read": (C.c_int, [_CTX, C.c_int, _P(C.c_double), _P(C.c_int64), C.c_int]),
    "livo2_map_upload": (C.c_int, [_CTX, _P(MapView)]),
    "livo2_map_update_planes": (C.c_int, [_CTX, _P(C.c_int32), C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_float), _P(C.c_float)]),
    "livo2_imu_propagate": (C.c_int, [_CTX, _P(State), C.c_void_p, C.c_int32, _P(ImuCfg), _P(State), C.c_void_p]),
    "livo2_imu_propagate_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_plane_fit_batch": (C.c_int, [_CTX, _P(C.c_double), _P(C.c_double), _P(C.c_int32), C.c_int32, C.c_float, _P(C.c_int32), _P(PlaneFit)]),
    "livo2_plane_fit_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_lidar_set_scan": (C.c_int, [_CTX, _P(C.c_float), C.c_int32, _P(LidarCfg)]),
    "livo2_lidar_preprocess_scan": (C.c_int, [_CTX, _P(C.c_float), _P(C.c_float), C.c_int32, C.c_void_p, C.c_int32, _P(C.c_double), _P(C.c_double), C.c_double,
                                              _P(LidarCfg), _P(C.c_int32), _P(C.c_float), _P(C.c_float)]),
    "livo2_lidar_preprocess_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_lidar_iterate": (C.c_int, [_CTX, _P(State), _P(State), _P(LidarCfg), _P(LidarSums), _P(LidarPoints)]),
    "livo2_lidar_update": (C.c_int, [_CTX, _P(State), _P(State), _P(LidarCfg), _P(LidarResult), _P(LidarPoints)]),
    "livo2_lidar_update_async": (C.c_int, [_CTX, _P(State), _P(State), _P(LidarCfg), _P(LidarPoints)]),
    "livo2_lidar_update_fetch": (C.c_int, [_CTX, _P(LidarResult), _P(LidarPoints)]),
    "livo2_lidar_iterations_async": (C.c_int, [_CTX, _P(State), _P(State), _P(LidarCfg), C.c_int32]),
    "livo2_lio_frame": (C.c_int, [_CTX, _P(State), C.c_void_p, C.c_int32, _P(ImuCfg), C.c_void_p, _P(C.c_float), _P(C.c_float), C.c_int32, C.c_double, _P(LidarCfg),
                                  _P(State), C.c_void_p, _P(C.c_int32), _P(LidarResult)]),
    "livo2_lidar_batch_set_scans": (C.c_int, [_CTX, C.c_int32, _P(C.c_float), _P(C.c_int32), _P(LidarCfg)]),
    "livo2_lidar_batch_update": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(LidarCfg), _P(LidarResult)]),
    "livo2_lidar_batch_update_async": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(LidarCfg)]),
    "livo2_lidar_batch_update_fetch": (C.c_int, [_CTX, C.c_int32, _P(LidarResult)]),
    "livo2_lidar_batch_iterations_async": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(LidarCfg), C.c_int32]),
    "livo2_visual_set_frame": (C.c_int, [_CTX, _P(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, _P(C.c_double), _P(C.c_float), _P(C.c_int32),
                                         _P(C.c_double), C.c_int32, C.c_int32]),
    "livo2_visual_set_reference": (C.c_int, [_CTX, _P(C.c_uint8), C.c_int32, _P(C.c_int32), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "livo2_visual_map_upload": (C.c_int, [_CTX, C.c_int32, _P(C.c_double), _P(C.c_int64), _P(C.c_uint8)]),
    "livo2_visual_select": (C.c_int, [_CTX, _P(C.c_double), C.c_int32, _P(SelectCfg), _P(C.c_int32), _P(C.c_float), _P(C.c_uint8), _P(C.c_uint8)]),
    "livo2_visual_select_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_visual_retrieve_warp": (C.c_int, [_CTX, _P(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, _P(C.c_uint8), C.c_int32, _P(RetrieveCandidates), _P(RetrieveCfg),
                                             _P(RetrieveOut), _P(C.c_int32)]),
    "livo2_visual_retrieve_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_visual_obs_upload": (C.c_int, [_CTX, _P(VisualObs)]),
    "livo2_frame_update_async": (C.c_int, [_CTX, _P(FrameIn)]),
    "livo2_frame_update_fetch": (C.c_int, [_CTX, _P(LidarResult), _P(VisualResult)]),
    "livo2_frame_update": (C.c_int, [_CTX, _P(FrameIn), _P(LidarResult), _P(VisualResult)]),
    "livo2_visual_map_apply": (C.c_int, [_CTX, _P(VisualMapDelta)]),
    "livo2_visual_map_counts": (C.c_int, [_CTX, _P(C.c_int32)]),
    "livo2_visual_retrieve_from_map": (C.c_int, [_CTX, _P(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, _P(C.c_double), C.c_int32, _P(SelectCfg), _P(RetrieveCfg),
                                                 _P(RetrieveChainOut), _P(C.c_int32), _P(C.c_int32)]),
    "livo2_visual_retrieve_from_map_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_visual_iterate": (C.c_int, [_CTX, C.c_int32, _P(State), _P(VisualCfg), _P(VisualSums), _P(C.c_float), _P(C.c_double), _P(C.c_double)]),
    "livo2_visual_update": (C.c_int, [_CTX, _P(State), _P(State), _P(VisualCfg), _P(VisualResult), _P(C.c_float)]),
    "livo2_visual_update_async": (C.c_int, [_CTX, _P(State), _P(State), _P(VisualCfg)]),
    "livo2_visual_update_fetch": (C.c_int, [_CTX, _P(VisualResult), _P(C.c_float)]),
    "livo2_visual_iterations_async": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(VisualCfg), C.c_int32]),
    "livo2_map_tree_create": (C.c_int, [_CTX, _P(MapTreeCfg)]),
    "livo2_map_tree_update": (C.c_int, [_CTX, _P(C.c_double), _P(C.c_double), C.c_int32, C.c_int32]),
    "livo2_map_tree_update_from_scan": (C.c_int, [_CTX, _P(State), _P(LidarCfg), C.c_int32]),
    "livo2_map_tree_update_from_scan_async": (C.c_int, [_CTX, _P(State), _P(LidarCfg)]),
    "livo2_map_tree_update_join": (C.c_int, [_CTX]),
    "livo2_map_tree_read_pv": (C.c_int, [_CTX, _P(C.c_double), _P(C.c_double), C.c_int32, _P(C.c_int32)]),
    "livo2_map_tree_stats": (C.c_int, [_CTX, _P(C.c_int32)]),
    "livo2_map_tree_slide": (C.c_int, [_CTX, _P(C.c_double), C.c_double, C.c_int32, _P(C.c_int32), _P(C.c_int32)]),
    "livo2_map_tree_export": (C.c_int, [_CTX, _P(C.c_int64), _P(C.c_int32), _P(C.c_double), _P(C.c_float), _P(C.c_int32), _P(C.c_int32), _P(C.c_double), _P(C.c_double),
                                        _P(C.c_double), _P(C.c_float), _P(C.c_float), _P(C.c_int32)]),
    "livo2_map_tree_read_planes": (C.c_int, [_CTX, _P(C.c_int32), C.c_int32, _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_float), _P(C.c_float), _P(C.c_int32)]),
    "livo2_map_tree_last_kernel_us": (C.c_double, [_CTX]),
    "livo2_visual_batch_set_references": (C.c_int, [_CTX, C.c_int32, _P(C.c_uint8), _P(C.c_int32), _P(C.c_int32), _P(C.c_double), _P(C.c_double), _P(C.c_double), _P(C.c_double)]),
    "livo2_visual_batch_set_frames": (C.c_int, [_CTX, C.c_int32, _P(C.c_uint8), C.c_int32, C.c_int32, C.c_int32, _P(C.c_double), _P(C.c_float), _P(C.c_int32), _P(C.c_double),
                                                _P(C.c_int32), C.c_int32]),
    "livo2_visual_batch_update": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(VisualCfg), _P(VisualResult)]),
    "livo2_visual_batch_update_async": (C.c_int, [_CTX, C.c_int32, _P(State), _P(State), _P(VisualCfg)]),
    "livo2_visual_batch_update_fetch": (C.c_int, [_CTX, C.c_int32, _P(VisualResult)]),
    "livo2_visual_batch_iterations_async": (C.c_int, [_CTX, C.c_int32, C.c_int32, _P(State), _P(State), _P(VisualCfg), C.c_int32]),
    "livo2_esikf_solve": (C.c_int, [_CTX, _P(C.c_double), _P(C.c_double), C.c_int32, C.c_double, C.c_int32, _P(State), _P(State), _P(State),
                                    _P(C.c_double), _P(C.c_double)]),
}

_lib = None


def load_library():
    """Load liblivo2_hip.so (in-tree).  Raises if it has not been built — never falls back to anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `make -C fast-livo2_amd/csrc` (or __graft_entry__.build()); "
                           "there is no CPU fallback for the ESIKF update path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def as_ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))
