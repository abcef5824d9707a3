// liblivo2_hip.so — C ABI implementation (include/livo2_hip.h) over the gfx950 kernels.
// Host side of the boundary: context / stream ownership, VoxelMap snapshot packing (open-addressing hash + 256-B plane
// records), scan and frame uploads, and the static launch sequences of the two ESIKF updates.  No CPU compute path exists
// here: every entry point either drives the HIP kernels or fails.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include "dev_alloc.hpp"
#include "lidar_kernels.hpp"
#include "visual_inverse_kernels.hpp"
#include "map_kernels.hpp"
#include "map_tree_kernels.hpp"
#include "retrieve_kernels.hpp"
#include "preprocess_kernels.hpp"
#include "select_kernels.hpp"
#include "raycast_kernels.hpp"
#include "choice_kernels.hpp"
#include "imu_kernels.hpp"
#include "frame_kernels.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

namespace {

struct HostIn { livo2_state cur, prop; DevHeader hdr; };      // pinned mirror of the head of DevCtl
#define IN_RING 16

struct EvPair { hipEvent_t a, b; };

struct TimingBin { std::vector<EvPair> used; double total_ms = 0; int64_t launches = 0; };

} // namespace

struct livo2_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  std::string err;
  DevCtl *d_ctl = nullptr;
  HostIn *h_in = nullptr;                   // pinned ring of IN_RING staging blocks: an update's H2D never waits for the previous update
  hipEvent_t in_ev[16] = {};                // recorded behind the H2D that reads slot k
  bool in_used[16] = {};
  int in_next = 0;
  void *h_out = nullptr;                    // pinned, sizeof(livo2_visual_result) (largest result)
  void *h_pts = nullptr; size_t h_pts_cap = 0;   // pinned staging of the per-point outputs
  void *scan_stage[2] = {nullptr, nullptr}; size_t scan_stage_cap[2] = {0, 0}; hipEvent_t scan_stage_ev[2] = {nullptr, nullptr}; bool scan_stage_used[2] = {false, false}; int scan_stage_next = 0;
  // map
  bool has_map = false;
  DevMap map{};
  RootSlot *d_slots = nullptr; double *d_cand = nullptr; double *d_planes = nullptr;      // slots; candidate hot words [.][16]; master plane records [.][32]
  double *d_planes_hot = nullptr; PlaneAux *d_plane_aux = nullptr, *d_cand_aux = nullptr;  // the residual kernel's view of the plane table (livo2_device.hpp)
  std::vector<int32_t> plane_cand_pos;      // host: position of each (caller-indexed) plane in the candidate array, or -1
  std::vector<int32_t> plane_internal, plane_orig;   // caller plane index <-> device (Morton-ordered) plane index
  int32_t *d_plane_internal = nullptr, *d_plane_cand_pos = nullptr; size_t plane_tab_cap = 0, plane_tab_cap2 = 0;   // device copies for k_plane_fit
  bool plane_tabs_fresh = false;
  // plane fit staging
  double *d_fit_pw = nullptr, *d_fit_var = nullptr; size_t fit_pw_cap = 0, fit_var_cap = 0;
  int32_t *d_fit_off = nullptr, *d_fit_idx = nullptr, *d_fit_list = nullptr; size_t fit_off_cap = 0, fit_idx_cap = 0, fit_list_cap = 0;
  livo2_plane_fit *d_fit_out = nullptr; size_t fit_out_cap = 0;
  double fit_kernel_us = 0.0;
  // retrieval candidates (N2)
  int cand_cap = 0; double *d_c_pos = nullptr, *d_c_normal = nullptr, *d_c_px = nullptr, *d_c_f = nullptr, *d_c_R = nullptr, *d_c_t = nullptr, *d_c_ie = nullptr, *d_c_ncc = nullptr, *d_c_A = nullptr;
  int32_t *d_c_idx = nullptr, *d_c_lvl = nullptr, *d_c_acc = nullptr, *d_c_sl = nullptr, *d_c_slot = nullptr, *d_c_count = nullptr; float *d_c_err = nullptr, *d_c_patch = nullptr; size_t c_patch_cap = 0;
  double retrieve_kernel_us = 0.0;
  // raw-scan pre-stage (N3)
  float *d_raw = nullptr, *d_curv = nullptr; size_t raw_cap = 0, curv_cap = 0; double *d_poses = nullptr; size_t poses_cap = 0;
  int32_t *d_vg_head = nullptr, *d_vg_slot = nullptr, *d_vg_misc = nullptr; size_t vg_head_cap = 0, vg_slot_cap = 0;
  double preprocess_kernel_us = 0.0;
  // visual map mirror + selection (N2)
  bool has_vmap = false; int n_vm = 0; double *d_vm_pos = nullptr; size_t vm_pos_cap = 0; unsigned long long *d_vm_pkey = nullptr; size_t vm_pkey_cap = 0;
  uint8_t *d_vm_active = nullptr, *d_vm_fov = nullptr; size_t vm_active_cap = 0, vm_fov_cap = 0;
  double *d_sel_pg = nullptr; size_t sel_pg_cap = 0; unsigned long long *d_sel_set = nullptr, *d_sel_depth = nullptr, *d_sel_best = nullptr; size_t sel_set_cap = 0, sel_depth_cap = 0, sel_best_cap = 0;
  int32_t *d_sel_type = nullptr, *d_sel_point = nullptr, *d_sel_flag = nullptr; float *d_sel_dist = nullptr; uint8_t *d_sel_disc = nullptr; size_t sel_type_cap = 0, sel_point_cap = 0, sel_dist_cap = 0, sel_disc_cap = 0;
  double select_kernel_us = 0.0;
  // RayCasting module (raycast_kernels.hpp): voxel set of the visual map, per-call ray state, add_from_voxel_map of the last call
  unsigned long long *d_vm_set = nullptr; size_t vm_set_cap = 0; uint32_t vm_set_mask = 0;
  unsigned long long *d_ray_set = nullptr, *d_ray_key = nullptr, *d_ray_hit_key = nullptr, *d_ray_hit_best = nullptr; size_t ray_set_cap = 0, ray_key_cap = 0, ray_hit_key_cap = 0, ray_hit_best_cap = 0;
  int32_t *d_ray_action = nullptr, *d_ray_hit_cell = nullptr, *d_ray_counters = nullptr; size_t ray_action_cap = 0, ray_hit_cell_cap = 0;
  double *d_ray_add = nullptr; size_t ray_add_cap = 0; int ray_add_n = -1;
  // observation table of the visual map, reference-patch choice, chained retrieval (N2)
  bool has_obs = false; int n_obs = 0, ob_n_ref = 0, ob_w = 0, ob_h = 0, ob_img_stride = 0;
  int32_t *d_ob_off = nullptr, *d_ob_id = nullptr, *d_ob_img = nullptr, *d_ob_lvl = nullptr, *d_vm_refpatch = nullptr;
  size_t ob_off_cap = 0, ob_id_cap = 0, ob_img_cap = 0, ob_lvl_cap = 0, vm_refpatch_cap = 0;
  double *d_ob_px = nullptr, *d_ob_f = nullptr, *d_ob_R = nullptr, *d_ob_t = nullptr, *d_ob_ie = nullptr, *d_vm_normal = nullptr;
  size_t ob_px_cap = 0, ob_f_cap = 0, ob_R_cap = 0, ob_t_cap = 0, ob_ie_cap = 0, vm_normal_cap = 0;
  float *d_ob_patch = nullptr; size_t ob_patch_cap = 0; uint8_t *d_vm_ninit = nullptr, *d_ob_imgs = nullptr; size_t vm_ninit_cap = 0, ob_imgs_cap = 0;
  // per-point observation lists (fixed stride, global observation indices in obs_ list order) + staging of livo2_visual_map_apply
  int32_t *d_ob_list = nullptr, *d_ob_cnt = nullptr; size_t ob_list_cap = 0, ob_cnt_cap = 0; int ob_stride = 0;
  void *h_delta = nullptr; size_t h_delta_cap = 0; void *d_delta = nullptr; size_t d_delta_cap = 0; hipEvent_t delta_ev = nullptr; bool delta_ev_used = false;
  int vm_delta_calls = 0, vm_delta_grows = 0;
  // whole frames (livo2_frame_update_*): pinned staging of the image + sub-map (two blocks, like the scan's), pinned ring of two result slots
  void *frame_stage[2] = {nullptr, nullptr}; size_t frame_stage_cap[2] = {0, 0}; hipEvent_t frame_stage_ev[2] = {nullptr, nullptr}; bool frame_stage_used[2] = {false, false}; int frame_stage_next = 0;
  void *h_frame_res = nullptr; hipEvent_t frame_res_ev[2] = {nullptr, nullptr}; int frame_head = 0, frame_inflight = 0;
  // head and tail of a frame as single launches (frame_kernels.hpp).  scan_small_fused: a scan of <= 16 384 points is prepared by two launches (keys with the frame's
  // input scatter; order by counting + gather + body covariance) instead of ~10; frame_ingest: 0 = one copy command per input array (round 4), 1 = one H2D into an arena + one scatter launch,
  // 2 = the launch reads the pinned staging block itself (payloads up to FRAME_ZERO_COPY_MAX bytes, larger ones go through the arena); frame_publish: the two result
  // blocks + the watchdog flag leave through one launch instead of three D2H copies.  Options of the same names / LIVO2_SCAN_SMALL_FUSED, LIVO2_FRAME_INGEST, LIVO2_FRAME_PUBLISH.
  int scan_small_fused = [] { const char *e = std::getenv("LIVO2_SCAN_SMALL_FUSED"); return e ? (std::atoi(e) != 0 ? 1 : 0) : 1; }();
  int frame_ingest = [] { const char *e = std::getenv("LIVO2_FRAME_INGEST"); const int v = e ? std::atoi(e) : 2; return (v >= 0 && v <= 2) ? v : 2; }();
  int frame_publish = [] { const char *e = std::getenv("LIVO2_FRAME_PUBLISH"); return e ? (std::atoi(e) != 0 ? 1 : 0) : 1; }();
  unsigned char *d_frame_arena = nullptr; size_t frame_arena_cap = 0;
  int scan_small_launches = 0, frame_ingest_launches = 0, frame_zero_copy_launches = 0, frame_publish_launches = 0;
  livo2_visual_cfg frame_vcfg[2] = {}; int frame_M[2] = {0, 0}; bool frame_persistent[2] = {false, false};
  bool vp_last_chained = false;
  int32_t *d_ch_obs = nullptr, *d_ch_flag = nullptr, *d_ch_slot = nullptr, *d_cand_cell = nullptr, *d_cand_point = nullptr, *d_cand_obs = nullptr, *d_sub_point = nullptr,
          *d_sub_obs = nullptr, *d_ch_count = nullptr;
  size_t ch_obs_cap = 0, ch_flag_cap = 0, ch_slot_cap = 0, cand_cell_cap = 0, cand_point_cap = 0, cand_obs_cap = 0, sub_point_cap = 0, sub_obs_cap = 0;
  int32_t *d_c_id = nullptr, *d_c_leader = nullptr, *d_ld_keys = nullptr, *d_ld_vals = nullptr; size_t c_id_cap = 0, c_leader_cap = 0, ld_keys_cap = 0, ld_vals_cap = 0;
  double chain_kernel_us = 0.0;
  int32_t *d_ret_blob = nullptr; size_t ret_blob_cap = 0; void *h_ret = nullptr; size_t h_ret_cap = 0;      // packed results of livo2_visual_retrieve_from_map (k_ret_pack), pinned landing block
  void *h_img = nullptr; size_t h_img_cap = 0;                                                              // pinned staging of that call's image
  hipStream_t stream_img = nullptr; hipEvent_t img_ready = nullptr;                                          // that image's DMA runs beside the selection kernels
  void *h_rp = nullptr; size_t h_rp_cap = 0;                                                                // pinned block of livo2_map_tree_read_planes (rows up, records down)
  void *h_err = nullptr; size_t h_err_cap = 0;                                                              // pinned landing block of livo2_visual_update_fetch's errors[]
  // IMU propagation (N4)
  double *d_imu_steps = nullptr, *d_imu_poses = nullptr; size_t imu_steps_cap = 0, imu_poses_cap = 0; livo2_state *d_imu_state = nullptr;   // [2]: in, out
  double imu_kernel_us = 0.0;
  // scan
  bool has_scan = false;
  int n = 0, n_cap = 0, lidar_block = 256;
  float *d_xyz_aos = nullptr, *d_x = nullptr, *d_y = nullptr, *d_z = nullptr; double *d_cb = nullptr;
  uint32_t *d_keys = nullptr, *d_keys2 = nullptr; int32_t *d_idx = nullptr, *d_perm = nullptr; void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
  unsigned long long *d_lidar_stamps = nullptr, *d_lidar_span_acc = nullptr; size_t lidar_stamps_cap = 0;      // device-clock duration of k_lidar_residual in a timed pass (k_lidar_span_acc)
  double *d_partials = nullptr; size_t partials_cap = 0;
  double *d_bcov_rows = nullptr; size_t bcov_rows_cap = 0;      // body_cov_list_ as 3x3 rows in caller order (fetch_lidar_points)
  int32_t *d_match = nullptr, *d_normal_plane = nullptr; float *d_dis = nullptr, *d_pw = nullptr; double *d_var = nullptr, *d_rinv = nullptr, *d_hrow = nullptr;
  int out_cap = 0;
  livo2_lidar_points want_l{};              // which per-point arrays the last enqueue produced
  // frame
  bool has_frame = false;
  uint8_t *d_img = nullptr; size_t img_cap = 0; int width = 0, height = 0, stride = 0;
  double *d_pos = nullptr, *d_invexpo = nullptr; float *d_warp = nullptr; int32_t *d_search = nullptr; int M = 0, L = 0, M_cap = 0; size_t warp_cap = 0;
  float *d_errors = nullptr; double *d_zdbg = nullptr, *d_Hdbg = nullptr; int dbg_cap = 0;
  // inverse-compositional reference data
  bool has_ref = false;
  uint8_t *d_ref_imgs = nullptr; size_t ref_img_cap = 0; int n_ref = 0;
  int32_t *d_ref_idx = nullptr; double *d_ref_px = nullptr, *d_ref_f = nullptr, *d_ref_R = nullptr, *d_ref_pos = nullptr, *d_gref = nullptr, *d_mref = nullptr; int ref_cap = 0;
  // batch of frames (independent scans + states against the resident map, one grid per ESIKF iteration)
  bool has_batch = false;
  int bn = 0;                                // frames in the batch
  std::vector<int32_t> b_count, b_off, b_grid, b_block_begin;   // per frame: points, first point, blocks, first block
  int b_total = 0, b_cap = 0, b_blocks = 0;
  float *bd_xyz_aos = nullptr, *bd_x = nullptr, *bd_y = nullptr, *bd_z = nullptr; double *bd_cb = nullptr;
  uint32_t *bd_keys = nullptr, *bd_keys2 = nullptr; int32_t *bd_idx = nullptr, *bd_perm = nullptr;
  double *bd_partials = nullptr; size_t b_partials_cap = 0;
  int32_t *bd_block_frame = nullptr; size_t b_block_frame_cap = 0;
  DevCtl *bd_ctl = nullptr; LidarBatchEntry *bd_entries = nullptr; HostIn *bd_in = nullptr; livo2_lidar_result *bd_results = nullptr;   // [LIVO2_MAX_BATCH]
  HostIn *bh_in = nullptr; livo2_lidar_result *bh_results = nullptr; LidarBatchEntry *bh_entries = nullptr;                              // pinned
  // batch of frames, visual (own images / sub-maps / states, lockstep (level, iteration) grids); shares bd_ctl / bd_in / bh_in with the LiDAR batch
  bool has_vbatch = false, has_vb_ref = false;
  uint8_t *vbd_ref_imgs = nullptr; int32_t *vbd_ref_idx = nullptr; double *vbd_ref_px = nullptr, *vbd_ref_f = nullptr, *vbd_ref_R = nullptr, *vbd_ref_pos = nullptr, *vbd_gref = nullptr, *vbd_mref = nullptr;
  size_t vb_ref_imgs_cap = 0, vb_ref_idx_cap = 0, vb_ref_px_cap = 0, vb_ref_f_cap = 0, vb_ref_R_cap = 0, vb_ref_pos_cap = 0, vb_gref_cap = 0, vb_mref_cap = 0;
  std::vector<int> vb_ref_cnt, vb_ref_off;      // reference images per frame of the batch / index of a frame's first one
  int vbn = 0, vb_total = 0, vb_blocks = 0, vb_L = 0, vb_w = 0, vb_h = 0, vb_stride = 0;
  std::vector<int32_t> vb_count, vb_off, vb_grid, vb_block_begin;
  uint8_t *vbd_img = nullptr; size_t vb_img_cap = 0;
  double *vbd_pos = nullptr, *vbd_invexpo = nullptr, *vbd_partials = nullptr; float *vbd_warp = nullptr, *vbd_errors = nullptr; int32_t *vbd_search = nullptr, *vbd_block_frame = nullptr;
  size_t vb_pos_cap = 0, vb_invexpo_cap = 0, vb_partials_cap = 0, vb_warp_cap = 0, vb_errors_cap = 0, vb_search_cap = 0, vb_block_frame_cap = 0;
  VisualBatchEntry *vbd_entries = nullptr, *vbh_entries = nullptr;      // [LIVO2_MAX_BATCH], device / pinned
  livo2_visual_result *vbd_results = nullptr, *vbh_results = nullptr;   // [LIVO2_MAX_BATCH], device / pinned
  // device-resident VoxelMap (map_tree_kernels.hpp)
  // persistent visual update (k_visual_update_persistent): one launch per computeJacobianAndUpdateEKF.  LIVO2_VISUAL_PERSISTENT=0 (or livo2_ctx_set_option) selects
  // the launch-per-step sequence instead.
  bool visual_persistent = [] { const char *e = std::getenv("LIVO2_VISUAL_PERSISTENT"); return e ? std::atoi(e) != 0 : true; }();
  bool visual_persistent_inverse = true;      // the inverse-compositional form on the resident grid too (option "visual_persistent_inverse")
  bool visual_error_waves = true;             // the frame error's float chains on groups of lanes (float_chain.hpp) when the threads' blocks are long enough (option "visual_error_waves")
  unsigned long long *d_vp_rows = nullptr; size_t vp_rows_cap = 0; unsigned long long *d_vp_errs = nullptr; size_t vp_errs_cap = 0;
  bool vp_xchg_dirty = true; size_t vp_err_pitch = 0; int vp_hw_rows = 0, vp_hw_m = 0;      // exchange buffers of k_visual_update_persistent: host-side view (api_visual.inc)
  // block order of k_lidar_residual (lidar_kernels.hpp, LptArgs): lifetimes per chunk written by every launch, order written by every solve; valid once a solve of this scan has run
  int32_t *d_lidar_tickets = nullptr;
  int32_t *d_lpt_order = nullptr; uint32_t *d_lpt_cost = nullptr; size_t lpt_order_cap = 0, lpt_cost_cap = 0; int lpt_chunks = 0; bool lpt_valid = false;
  bool lidar_block_order = [] { const char *e = std::getenv("LIVO2_LIDAR_BLOCK_ORDER"); return e ? std::atoi(e) != 0 : true; }();
  // one launch per ESIKF iteration (k_lidar_iteration: the last block of the residual grid to arrive reduces and solves) instead of k_lidar_residual + k_lidar_solve:
  // option "lidar_fused_iteration" / LIVO2_LIDAR_FUSED=1.  Same results bit for bit (tests/test_bench_workload_gpu.py) — and measured SLOWER at C4 (31.0 against 28.2 us
  // per iteration, profiles/r05_lidar_fused_iteration_ab.txt: every block pays a store drain + an atomic round trip before it may leave), so the default stays off.
  bool lidar_fused = [] { const char *e = std::getenv("LIVO2_LIDAR_FUSED"); return e ? std::atoi(e) != 0 : false; }();
  int lidar_fused_launches = 0;
  hipEvent_t vp_done = nullptr; int vp_blocks_inflight = 0;     // this ctx's last persistent launch (device-wide accounting below)
  unsigned long long *d_vp_prof = nullptr; bool vp_prof = [] { const char *e = std::getenv("LIVO2_VP_PROF"); return e ? std::atoi(e) != 0 : false; }();
  int vp_used = 0, vp_fallback = 0, vp_timeouts = 0;             // statistics: persistent launches / fallbacks to the per-step sequence / grids that gave up and were re-run per step
  bool vp_debug_timeout = false, vp_rerun = false, vp_last_valid = false;
  // watchdog of the resident grid: a word that does not arrive within vp_timeout_us makes every block leave; the update is then re-run per step.  Default 20 ms
  // (a C4 update is 0.25 ms; a 10 Hz pipeline must not spin for seconds): option "visual_persistent_timeout_us" / LIVO2_VP_TIMEOUT_US.  After a time-out the ctx
  // uses the launch-per-step sequence for vp_backoff_left further updates (8, doubling up to 1024 while time-outs repeat) before it tries a resident grid again.
  int vp_timeout_us = [] { const char *e = std::getenv("LIVO2_VP_TIMEOUT_US"); const int v = e ? std::atoi(e) : 0; return v > 0 ? v : 20000; }();
  int vp_backoff_left = 0, vp_backoff_len = 0, vp_backoff_skips = 0;
  livo2_state vp_last_in{}, vp_last_prop{}; livo2_visual_cfg vp_last_cfg{};       // inputs of the last persistent launch (a timed-out grid is re-run from them)
  bool tree_mode = false;
  MapTreeArgs mt{};
  double mt_last_slide[3] = {0, 0, 0};      // VoxelMapManager::last_slide_position
  bool mt_pool_pressure = false;            // some pool of the device tree is more than half used (or LIVO2_MAP_RECYCLE=1): updates run the recycling kernels
  livo2_map_tree_cfg mt_cfg{};
  double *mt_in_pw = nullptr, *mt_in_var = nullptr; size_t mt_in_pw_cap = 0, mt_in_var_cap = 0;
  unsigned long long *mt_keys = nullptr, *mt_keys2 = nullptr; size_t mt_keys_cap = 0, mt_keys2_cap = 0;
  int32_t *mt_idx = nullptr, *mt_order = nullptr, *mt_head = nullptr, *mt_slot = nullptr, *mt_seg_begin = nullptr, *mt_seg_root = nullptr, *mt_nseg = nullptr;
  size_t mt_idx_cap = 0, mt_order_cap = 0, mt_head_cap = 0, mt_slot_cap = 0, mt_seg_begin_cap = 0, mt_seg_root_cap = 0;
  livo2_state *mt_state = nullptr;
  int mt_spread_opt = 0, mt_wide_fit_opt = 1;    // options "map_update_spread" (0: by the previous update's touched roots; 1 / 2 / 4 / 8) and "map_update_wide_fit"
  int mt_last_touched = 0;                  // root voxels the last update touched (sizes the next one's lanes per root)
  int mt_pv_n = -1;                         // points of the pv_list in mt_in_pw / mt_in_var (last map-tree update), -1: none since the last set_scan
  int32_t *mt_rp_rows = nullptr; size_t mt_rp_rows_cap = 0; double *mt_rp_out = nullptr; size_t mt_rp_out_cap = 0;   // livo2_map_tree_read_planes staging
  double mt_kernel_us = 0.0;
  // livo2_map_tree_update_from_scan_async: the octree update on a second stream of the context, concurrently with the visual half of the frame (api_map_tree.inc)
  hipStream_t stream_mt = nullptr; hipEvent_t mt_fork = nullptr, mt_ev0 = nullptr, mt_ev1 = nullptr; bool mt_async = false;
  void *mt_sort_tmp = nullptr; size_t mt_sort_tmp_bytes = 0;
  int32_t *h_mt_cnt = nullptr;               // pinned: the pool counters of an update on the second stream, copied behind its last kernel (the join reads them without a round trip)
  // the ~25 launches of such an update are enqueued by a helper thread of the context: on the caller's thread they stood (0.13 ms) between the LiDAR update and the
  // retrieval they are meant to run beside
  std::thread mt_worker; std::mutex mt_mu; std::condition_variable mt_cv; std::function<int()> mt_job; bool mt_job_pending = false, mt_job_done = true, mt_quit = false; int mt_job_rc = 0;
  std::string mt_job_err;
  int mt_grow_events = 0;                   // pool growths / candidate re-packs so far (livo2_ctx_get_counter "map_tree_grow_events")
#ifdef LIVO2_PHASE_PROF
  unsigned long long *d_prof = nullptr; size_t prof_waves = 0;
#endif
  // timing
  hipEvent_t span0 = nullptr, span1 = nullptr;   // bracket the kernels of one synchronous call (the *_last_kernel_us queries)
  bool timing = false;
  TimingBin bins[4];
  std::vector<EvPair> ev_pool;
};

namespace {

// where an error text goes: ctx->err, except on the context's helper thread (api_map_tree.inc), which keeps its own until the join hands it over
thread_local std::string *tl_err_sink = nullptr;
inline std::string &err_of(livo2_ctx *ctx) { return tl_err_sink ? *tl_err_sink : ctx->err; }
#define HIPCHK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      err_of(ctx) = std::string(#call) + ": " + hipGetErrorString(e_);                                      \
      return LIVO2_ERR_HIP;                                                                                 \
    }                                                                                                       \
  } while (0)

int fail(livo2_ctx *ctx, int code, const char *msg) { if (ctx) err_of(ctx) = msg; return code; }

// LIVO2_REDZONE=1 (dev_alloc.hpp): every synchronising entry point ends by scanning the guards of all device allocations of the process
int rz_gate(livo2_ctx *ctx) {
  if (devalloc::mode() != 1) return LIVO2_OK;
  char msg[320];
  const long long bad = devalloc::check(msg, sizeof(msg));
  if (bad == 0) return LIVO2_OK;
  ctx->err = bad < 0 ? "redzone check could not run" : msg;
  fprintf(stderr, "liblivo2_hip: %s\n", ctx->err.c_str());
  return LIVO2_ERR_HIP;
}

// (ensure / grow_array are macros over *_at so that the debug allocator records the CALLER's line: one line per buffer, dev_alloc.hpp)
template <typename T> int ensure_at(int line, livo2_ctx *ctx, T *&p, size_t &cap, size_t need) {
  if (need <= cap && p) return LIVO2_OK;
  if (p) { hipError_t e = DFREE(p); (void)e; p = nullptr; }
  size_t newcap = std::max(need, cap + cap / 2);
  HIPCHK(DMALLOC_AT(line, (void **)&p, newcap * sizeof(T)));
  cap = newcap;
  return LIVO2_OK;
}
#define ensure(...) ensure_at(LIVO2_HERE, __VA_ARGS__)

template <typename T> int grow_array_at(int line, livo2_ctx *ctx, T *&p, size_t old_n, size_t new_n, bool zero_tail) {
  T *q = nullptr;
  HIPCHK(DMALLOC_AT(line, (void **)&q, new_n * sizeof(T)));
  if (old_n) HIPCHK(devalloc::memcpy_async(q, p, old_n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  if (zero_tail && new_n > old_n) HIPCHK(hipMemsetAsync(q + old_n, 0, (new_n - old_n) * sizeof(T), ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(DFREE(p));
  p = q;
  return LIVO2_OK;
}
#define grow_array(...) grow_array_at(LIVO2_HERE, __VA_ARGS__)

// capacity for `need` elements, CONTENT KEPT (the first `used` elements): geometric growth, so a map that grows by a few points per frame re-allocates O(log) times
template <typename T> int keep_grow_at(int line, livo2_ctx *ctx, T *&p, size_t &cap, size_t used, size_t need) {
  if (need <= cap && p) return LIVO2_OK;
  const size_t newcap = std::max(need, 2 * cap + 64);
  T *q = nullptr;
  HIPCHK(DMALLOC_AT(line, (void **)&q, newcap * sizeof(T)));
  if (p && used) HIPCHK(devalloc::memcpy_async(q, p, used * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (p) HIPCHK(DFREE(p));
  p = q; cap = newcap;
  ctx->vm_delta_grows++;
  return LIVO2_OK;
}
#define keep_grow(...) keep_grow_at(LIVO2_HERE, __VA_ARGS__)
struct Blob {                                           // sections of the staging block, 16-byte aligned
  size_t size = 0;
  size_t add(size_t bytes) { const size_t at = size; size += (bytes + 15) & ~(size_t)15; return at; }
};

struct Timed {                 // RAII-free helper: brackets one launch with an event pair when timing is on
  livo2_ctx *ctx; int bin; EvPair ev{}; bool on = false;
  Timed(livo2_ctx *c, int b) : ctx(c), bin(b) {
    if (!c->timing) return;
    if (c->bins[b].used.size() >= 65536) return;
    if (c->ev_pool.empty()) { if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) return; }
    else { ev = c->ev_pool.back(); c->ev_pool.pop_back(); }
    on = true;
    hipError_t e = hipEventRecord(ev.a, c->stream); (void)e;
  }
  void done() { if (!on) return; hipError_t e = hipEventRecord(ev.b, ctx->stream); (void)e; ctx->bins[bin].used.push_back(ev); }
};

// (re)point the residual kernel's map view at the ctx's arrays
void set_map_view(livo2_ctx *ctx) {
  ctx->map.slots = ctx->d_slots; ctx->map.cand_rec = ctx->d_cand; ctx->map.cand_aux = ctx->d_cand_aux; ctx->map.planes = ctx->d_planes_hot; ctx->map.plane_aux = ctx->d_plane_aux;
}
void free_map_arrays(livo2_ctx *ctx) {
  void *p[] = {ctx->d_slots, ctx->d_cand, ctx->d_planes, ctx->d_planes_hot, ctx->d_plane_aux, ctx->d_cand_aux};
  for (void *q : p) if (q) { hipError_t e = DFREE(q); (void)e; }
  ctx->d_slots = nullptr; ctx->d_cand = nullptr; ctx->d_planes = nullptr; ctx->d_planes_hot = nullptr; ctx->d_plane_aux = nullptr; ctx->d_cand_aux = nullptr;
}

void pack_plane(double *rec, const double *normal, const double *center, const double *pv36, float d, float radius) {
  for (int k = 0; k < 3; k++) { rec[k] = normal[k]; rec[3 + k] = center[k]; }
  int q = 6;
  for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) rec[q++] = 0.5 * (pv36[a * 6 + b] + pv36[b * 6 + a]);   // J S J^T only sees sym(S)
  float dr[2] = {d, radius};
  std::memcpy(&rec[27], dr, 8);
  for (int k = 28; k < 32; k++) rec[k] = 0.0;
}

__global__ void k_scatter_planes(const double *__restrict__ recs, const int32_t *__restrict__ idx, int n, double *__restrict__ planes) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int p = t >> 5, k = t & 31;
  if (p >= n) return;
  planes[(size_t)idx[p] * PLANE_REC_DOUBLES + k] = recs[(size_t)p * PLANE_REC_DOUBLES + k];
}
// master records -> the residual kernel's view (hot words + side word), for the rows listed (rows == null: rows 0..n-1); gate_pos (optional): the
// plane's copy inside a candidate list is refreshed as well (its meta stays)
__global__ void k_planes_hot(const double *__restrict__ planes, const int32_t *__restrict__ rows, const int32_t *__restrict__ gate_pos, int n, double *__restrict__ hot_out,
                             PlaneAux *__restrict__ aux, double *__restrict__ cand, PlaneAux *__restrict__ cand_aux) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int row = rows ? rows[p] : p;
  const double *rec = planes + (size_t)row * PLANE_REC_DOUBLES;
  double f[28];
#pragma unroll
  for (int k = 0; k < 28; k++) f[k] = rec[k];
  double hot[PLANE_HOT_DOUBLES];
  plane_hot_words(f, f + 3, f + 6, hot);
  const float2 dr = __builtin_bit_cast(float2, f[27]);
  double2 *dst = reinterpret_cast<double2 *>(hot_out + (size_t)row * PLANE_HOT_DOUBLES);
#pragma unroll
  for (int k = 0; k < PLANE_HOT_DOUBLES / 2; k++) dst[k] = make_double2(hot[2 * k], hot[2 * k + 1]);
  PlaneAux x; x.d = dr.x; x.radius = dr.y; x.meta = row; x.pad = 0;
  aux[row] = x;
  const int gp = gate_pos ? gate_pos[p] : -1;
  if (gp >= 0) {
    double2 *cd = reinterpret_cast<double2 *>(cand + (size_t)gp * PLANE_HOT_DOUBLES);
#pragma unroll
    for (int k = 0; k < PLANE_HOT_DOUBLES / 2; k++) cd[k] = make_double2(hot[2 * k], hot[2 * k + 1]);
    cand_aux[gp].d = dr.x; cand_aux[gp].radius = dr.y;
  }
}
// candidate lists: hot words of plane (meta & mask), side word with the list's meta (plane | layer << 28)
__global__ void k_cand_fill(const double *__restrict__ hot, const PlaneAux *__restrict__ aux, const int32_t *__restrict__ meta, int n, double *__restrict__ cand, PlaneAux *__restrict__ cand_aux) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x, c = t >> 3, k = t & 7;
  if (c >= n) return;
  const int pl = meta[c] & CAND_PLANE_MASK;
  reinterpret_cast<double2 *>(cand + (size_t)c * PLANE_HOT_DOUBLES)[k] = reinterpret_cast<const double2 *>(hot + (size_t)pl * PLANE_HOT_DOUBLES)[k];
  if (k == 0) { PlaneAux x = aux[pl]; x.meta = meta[c]; cand_aux[c] = x; }
}

// head of DevCtl from a state that is already on the device (livo2_lio_frame: state_ = state_propagat = the IMU propagation's result): what
// upload_states() sends from the host — cur, prop, a cleared header, RE = rot_end * extR with the same expression
struct Mat9 { double v[9]; };
__global__ void __launch_bounds__(256) k_ctl_from_state(DevCtl *__restrict__ ctl, const livo2_state *__restrict__ st, Mat9 extR) {
  const int t = threadIdx.x;
  const double *src = reinterpret_cast<const double *>(st);
  double *c = reinterpret_cast<double *>(&ctl->cur), *p = reinterpret_cast<double *>(&ctl->prop);
  for (int k = t; k < (int)(sizeof(livo2_state) / 8); k += 256) { const double v = src[k]; c[k] = v; p[k] = v; }
  if (t == 0) { ctl->hdr.stop = 0; ctl->hdr.rematch_num = 0; ctl->hdr.reserved = 0; ctl->hdr.last_error = FLT_MAX; ctl->hdr.n_steps = 0; ctl->hdr.pad[0] = ctl->hdr.pad[1] = ctl->hdr.pad[2] = 0; }
  if (t < 9) { const int i = t / 3, j = t % 3; ctl->hdr.RE[t] = (st->rot[i * 3] * extR.v[j] + st->rot[i * 3 + 1] * extR.v[3 + j]) + st->rot[i * 3 + 2] * extR.v[6 + j]; }
}

__global__ void __launch_bounds__(LIVO2_WAVE) k_esikf_solve_only(DevCtl *__restrict__ ctl, int k, double scale, int sign) {
  __shared__ SolveLds s;
  const int lane = threadIdx.x;
  if (lane < k * k) s.hth[lane] = ctl->solve_hth[lane];
  if (lane < k) s.htz[lane] = ctl->solve_htz[lane];
  __syncthreads();
  double craw[6];
  esikf_prefetch_wave(ctl, s, scale, lane, craw);
  if (lane == 0) esikf_log_lane(ctl, s);
  __syncthreads();
  if (k == 6) esikf_update_wave<6>(ctl, s, sign, lane); else esikf_update_wave<7>(ctl, s, sign, lane);
  if (lane < DS) ctl->solve_solution[lane] = s.sol[lane];
}

#ifdef LIVO2_PHASE_PROF
#define SOLVE_PROF_ARG , (ctx->d_prof ? ctx->d_prof + ctx->prof_waves * 8 : nullptr)
#else
#define SOLVE_PROF_ARG
#endif
int lidar_grid(int n, int block) { int chunks = (n + block - 1) / block; int per_xcd = (chunks + 7) / 8; return std::max(8, per_xcd * 8); }
// Threads (= points) per block of a single-scan launch: 256.  128-point blocks are compiled in and selectable (LIVO2_LIDAR_BLOCK=128, tools/block_probe.py): at C4
// (200 000 points = 782 blocks of 256 against the 512 the chip holds at once) they shorten the residual kernel (25.4 -> 23.0 us) by the amount the solve, which
// then reads twice as many partial rows through ONE CU (~130 GB/s: 400 KB = 2.9 us), gets longer (10.3 -> 13.2 us) — no gain per iteration, so 256 stays the rule.
int lidar_block_for(int n) {
  static const int forced = [] { const char *e = std::getenv("LIVO2_LIDAR_BLOCK"); const int v = e ? std::atoi(e) : 0; return (v == 128 || v == 256) ? v : 0; }();
  (void)n;
  return forced ? forced : 256;
}
void launch_lidar_residual(livo2_ctx *ctx, const LidarKernelArgs &a, int check_stop);
// The launch order only matters when a scan needs more than one round of blocks (2 waves per SIMD: 8 waves = two 256-thread blocks per CU), and it costs every
// block a dependent scalar load at its start and a store at its end (measured: +0.6 us on a single-round launch): on above that size only.
bool lidar_lpt_on(livo2_ctx *ctx, int chunks) {
  static int cus[64] = {};
  if (!ctx->lidar_block_order || ctx->lpt_chunks != chunks || chunks > LPT_MAX_CHUNKS || ctx->device < 0 || ctx->device >= 64) return false;
  if (cus[ctx->device] == 0) { hipDeviceProp_t prop; cus[ctx->device] = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess ? prop.multiProcessorCount : 256; }
  return chunks * (ctx->lidar_block / LIVO2_WAVE) > cus[ctx->device] * 8;
}

void launch_lidar_residual(livo2_ctx *ctx, const LidarKernelArgs &a, int check_stop) {
  const int chunks = lidar_grid(std::max(ctx->n, 1), ctx->lidar_block);
  // LIVO2_LIDAR_RESIDENT=<blocks>: a resident grid looping over the chunks instead of one block per chunk (default).  Measured at C4 (782 chunks): 512 resident
  // blocks 26.5 us, 384: 26.4, 640: 24.5 against 24.7 us for one block per chunk — a static chunk -> block assignment loses more to the cluttered chunks than
  // it saves in block start-up; the hardware dispatcher's dynamic placement stays (profiles/r03_lidar_resident_grid_probe.txt).
  static const int resident = [] { const char *e = std::getenv("LIVO2_LIDAR_RESIDENT"); const int v = e ? std::atoi(e) : 0; return v > 0 ? (v + 7) / 8 * 8 : 0; }();
  const int grid = resident > 0 ? std::min(chunks, resident * (ctx->lidar_block == 128 ? 2 : 1)) : chunks;
  const bool lpt = lidar_lpt_on(ctx, chunks) && resident == 0;
  const int32_t *order = (lpt && ctx->lpt_valid) ? ctx->d_lpt_order : nullptr;
  uint32_t *cost = lpt ? ctx->d_lpt_cost : nullptr;
  unsigned long long *stamps = (ctx->timing && ctx->d_lidar_stamps && ctx->lidar_stamps_cap >= 2 * (size_t)chunks) ? ctx->d_lidar_stamps : nullptr;
  if (resident > 0) {
    if (ctx->lidar_block == 128) hipLaunchKernelGGL(k_lidar_residual_resident<128>, dim3(grid), dim3(128), LIDAR_LDS_BYTES_OF(128) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
    else hipLaunchKernelGGL(k_lidar_residual_resident<256>, dim3(grid), dim3(256), LIDAR_LDS_BYTES_OF(256) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
  } else if (ctx->lidar_block == 128) hipLaunchKernelGGL(k_lidar_residual<128>, dim3(chunks), dim3(128), LIDAR_LDS_BYTES_OF(128) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost, stamps);
  else hipLaunchKernelGGL(k_lidar_residual<256>, dim3(chunks), dim3(256), LIDAR_LDS_BYTES_OF(256) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost, stamps);
}

// One ESIKF iteration as ONE launch (lidar_kernels.hpp, k_lidar_iteration).  False: this configuration runs the two-launch sequence (128-point blocks, the resident-grid
// experiment, the profiling build with its stamps in k_lidar_solve, option off).
bool lidar_fused_on(livo2_ctx *ctx) {
#ifdef LIVO2_PHASE_PROF
  return false;
#else
  static const bool resident = [] { const char *e = std::getenv("LIVO2_LIDAR_RESIDENT"); return e && std::atoi(e) > 0; }();
  return ctx->lidar_fused && ctx->lidar_block == 256 && !resident;
#endif
}
void launch_lidar_iteration(livo2_ctx *ctx, const LidarKernelArgs &a, int check_stop, int mode, int iter, int max_iter) {
  const int chunks = lidar_grid(std::max(ctx->n, 1), ctx->lidar_block);
  const bool lpt = lidar_lpt_on(ctx, chunks);
  const int32_t *order = (lpt && ctx->lpt_valid) ? ctx->d_lpt_order : nullptr;
  uint32_t *cost = lpt ? ctx->d_lpt_cost : nullptr;
  const LidarFuseArgs fz = {mode, iter, max_iter, (int32_t)(ctx->lpt_order_cap / 2), cost, lpt ? ctx->d_lpt_order : nullptr, ctx->d_lidar_tickets};
  hipLaunchKernelGGL(k_lidar_iteration<256>, dim3(chunks), dim3(256), LIDAR_LDS_BYTES_OF(256) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost, fz);
  if (lpt) ctx->lpt_valid = true;                                  // the launches enqueued from here on read the order this one writes
  ctx->lidar_fused_launches++;
}

int check_lidar_cfg(livo2_ctx *ctx, const livo2_lidar_cfg *cfg) {
  if (!cfg) return fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL");
  if (cfg->max_iterations < 1 || cfg->max_iterations > LIVO2_MAX_ITERS) return fail(ctx, LIVO2_ERR_INVALID, "max_iterations out of [1,LIVO2_MAX_ITERS]");
  if (cfg->max_layer < 0 || cfg->max_layer > LIVO2_MAX_LAYER) return fail(ctx, LIVO2_ERR_INVALID, "max_layer out of [0,LIVO2_MAX_LAYER]");
  if (!(cfg->voxel_size > 0)) return fail(ctx, LIVO2_ERR_INVALID, "voxel_size must be > 0");
  // The first plane that passes the 3-sigma gate is taken without evaluating its probability (lidar_kernels.hpp, struct Best): exp(-d^2 / 2 sigma) / sqrt(sigma) > 0
  // whenever d < sigma_num sqrt(sigma) and sigma_num is moderate.  Beyond ~38 the exponential underflows, `this_prob > prob` (voxel_map.cpp:741) fails and the
  // reference pushes a default-constructed PointToPlane — a case nobody can mean; it is refused instead of reproduced.
  if (!(cfg->sigma_num > 0) || cfg->sigma_num > 30.0) return fail(ctx, LIVO2_ERR_INVALID, "sigma_num out of (0, 30]");
  return LIVO2_OK;
}

int upload_states(livo2_ctx *ctx, const livo2_state *cur, const livo2_state *prop, const double *extR = nullptr) {
  // pinned staging ring: slot k is rewritten only after the H2D that read it last has completed (an event per slot), so a caller that
  // enqueues update after update (livo2_*_update_async) never blocks here until IN_RING updates are in flight
  const int k = ctx->in_next;
  ctx->in_next = (k + 1) % IN_RING;
  if (ctx->in_used[k]) HIPCHK(hipEventSynchronize(ctx->in_ev[k]));
  HostIn *h = ctx->h_in + k;
  h->cur = *cur; h->prop = *prop;
  std::memset(&h->hdr, 0, sizeof(DevHeader));
  h->hdr.last_error = FLT_MAX;
  if (extR) {            // state_propagat.rot_end * extR_ (voxel_map.cpp:445) is constant during one update
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      h->hdr.RE[i * 3 + j] = (prop->rot[i * 3] * extR[j] + prop->rot[i * 3 + 1] * extR[3 + j]) + prop->rot[i * 3 + 2] * extR[6 + j];
  }
  HIPCHK(devalloc::memcpy_async(ctx->d_ctl, h, sizeof(HostIn), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->in_ev[k], ctx->stream));
  ctx->in_used[k] = true;
  return LIVO2_OK;
}

int ensure_lidar_outputs(livo2_ctx *ctx, const livo2_lidar_points *want) {
  ctx->want_l = livo2_lidar_points{};
  if (!want) return LIVO2_OK;
  ctx->want_l = *want;
  if (ctx->out_cap < ctx->n) {
    hipError_t e;
    if (ctx->d_match) { e = DFREE(ctx->d_match); e = DFREE(ctx->d_normal_plane); e = DFREE(ctx->d_dis); e = DFREE(ctx->d_pw); e = DFREE(ctx->d_var); e = DFREE(ctx->d_rinv); e = DFREE(ctx->d_hrow); (void)e; }
    size_t n = (size_t)ctx->n_cap;
    HIPCHK(DMALLOC((void **)&ctx->d_match, n * 4)); HIPCHK(DMALLOC((void **)&ctx->d_normal_plane, n * 4));
    HIPCHK(DMALLOC((void **)&ctx->d_dis, n * 4)); HIPCHK(DMALLOC((void **)&ctx->d_pw, n * 12));
    HIPCHK(DMALLOC((void **)&ctx->d_var, n * 72)); HIPCHK(DMALLOC((void **)&ctx->d_rinv, n * 8)); HIPCHK(DMALLOC((void **)&ctx->d_hrow, n * 48));
    ctx->out_cap = ctx->n_cap;
  }
  return LIVO2_OK;
}

// launch constants derived from voxel_size / sigma_num (lidar_kernels.hpp, LidarKernelArgs)
void lidar_args_consts(LidarKernelArgs &a) {
  int ex = 0;
  const double inv = 1.0 / a.voxel_size;
  a.inv_voxel_size = (std::frexp(a.voxel_size, &ex) == 0.5 && std::isfinite(inv) && inv >= 0x1p-500 && inv <= 0x1p500) ? inv : 0.0;      // a power of two: x * inv == x / voxel_size bit for bit
  const double sn2 = a.sigma_num * a.sigma_num;
  a.sn2_lo = sn2 * (1.0 - 1e-14); a.sn2_hi = sn2 * (1.0 + 1e-14);
}
LidarKernelArgs make_lidar_args(livo2_ctx *ctx, const livo2_lidar_cfg *cfg) {
  LidarKernelArgs a{};
  a.x = ctx->d_x; a.y = ctx->d_y; a.z = ctx->d_z; a.cb = ctx->d_cb; a.perm = ctx->d_perm; a.n = ctx->n; a.max_layer = cfg->max_layer; a.map = ctx->map;
  a.voxel_size = cfg->voxel_size; a.sigma_num = cfg->sigma_num;
  lidar_args_consts(a);
  std::memcpy(a.ER, cfg->extR, 72); std::memcpy(a.Et, cfg->extT, 24);
#ifdef LIVO2_PHASE_PROF
  {
    size_t waves = (size_t)lidar_grid(std::max(ctx->n, 1), 128) * 4;
    if (waves > ctx->prof_waves) { if (ctx->d_prof) { hipError_t e = DFREE(ctx->d_prof); (void)e; } hipError_t e = DMALLOC((void **)&ctx->d_prof, waves * 64 + 128 + waves * 16 + waves * 128); (void)e; ctx->prof_waves = waves; }
    hipError_t e = hipMemsetAsync(ctx->d_prof, 0, waves * 64, ctx->stream); (void)e;
    a.prof = ctx->d_prof;
  }
#endif
  const livo2_lidar_points &w = ctx->want_l;
  a.match_plane = w.match_plane ? ctx->d_match : nullptr; a.dis = w.dis_to_plane ? ctx->d_dis : nullptr; a.pw = w.point_w ? ctx->d_pw : nullptr;
  a.normal_plane = w.normal_plane ? ctx->d_normal_plane : nullptr; a.var = w.var ? ctx->d_var : nullptr;
  a.r_inv = w.r_inv ? ctx->d_rinv : nullptr; a.h_row = w.h_row ? ctx->d_hrow : nullptr;
  return a;
}

// Per-point outputs (pv_list_ / ptpl_list_ / body_cov_list_ members) come back through ONE pinned staging block owned by the ctx: every selected
// array is copied D2H into it asynchronously (pinned memory: DMA at link rate, no hidden pageable bounce buffer per call), one synchronisation, then a host copy
// into the caller's arrays.
// body_cov_list_ for the caller: the scan keeps the symmetric six of calcBodyCov (voxel_map.cpp:15-34) in sorted order; this writes full 3x3 matrices in the caller's order
__global__ void k_body_cov_rows(const double *__restrict__ cb, const int32_t *__restrict__ perm, int n, double *__restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / 9, e = t % 9;
  if (j >= n) return;
  const int r = e / 3, c = e % 3, u = r < c ? r : c, v = r < c ? c : r;
  out[(size_t)perm[j] * 9 + e] = cb[(size_t)(u * 3 - (u * (u - 1)) / 2 + (v - u)) * n + j];
}

int fetch_lidar_points(livo2_ctx *ctx, const livo2_lidar_points *p) {
  if (!p) return LIVO2_OK;
  const livo2_lidar_points &w = ctx->want_l;
  const size_t n = (size_t)ctx->n;
  if ((p->match_plane && !w.match_plane) || (p->dis_to_plane && !w.dis_to_plane) || (p->point_w && !w.point_w) || (p->normal_plane && !w.normal_plane) ||
      (p->var && !w.var) || (p->r_inv && !w.r_inv) || (p->h_row && !w.h_row))
    return fail(ctx, LIVO2_ERR_INVALID, "per-point array requested at fetch was not selected at enqueue");
  struct Item { void *dst; const void *src; size_t bytes; };
  if (p->body_cov && n > 0) {
    int rc = ensure(ctx, ctx->d_bcov_rows, ctx->bcov_rows_cap, n * 9); if (rc) return rc;
    hipLaunchKernelGGL(k_body_cov_rows, dim3((unsigned)((n * 9 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_cb, ctx->d_perm, (int)n, ctx->d_bcov_rows);
  }
  const Item items[] = {{p->match_plane, ctx->d_match, n * 4}, {p->dis_to_plane, ctx->d_dis, n * 4}, {p->point_w, ctx->d_pw, n * 12}, {p->normal_plane, ctx->d_normal_plane, n * 4},
                        {p->var, ctx->d_var, n * 72}, {p->r_inv, ctx->d_rinv, n * 8}, {p->h_row, ctx->d_hrow, n * 48}, {p->body_cov, ctx->d_bcov_rows, n * 72}};
  constexpr int NITEMS = 8;
  size_t total = 0;
  for (const Item &it : items) if (it.dst) total += (it.bytes + 63) & ~(size_t)63;
  if (total == 0) return LIVO2_OK;
  if (p->pinned || total > ((size_t)8 << 20)) {
    // page-locked destinations (p->pinned), or large outputs (C4: 168 B x 200 000 points = 34 MB): copy straight into the caller's arrays — the runtime pipelines a pageable D2H through its own pinned
    // chunks while it copies the previous chunk out, which a stage-everything-then-memcpy scheme does not (measured: 3.0 ms against 3.8 ms per C4 frame)
    for (int k = 0; k < NITEMS; k++) if (items[k].dst && items[k].bytes) HIPCHK(devalloc::memcpy_async(items[k].dst, items[k].src, items[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (!ctx->tree_mode) {
      if (p->match_plane) for (size_t i = 0; i < n; i++) if (p->match_plane[i] >= 0) p->match_plane[i] = ctx->plane_orig[p->match_plane[i]];
      if (p->normal_plane) for (size_t i = 0; i < n; i++) if (p->normal_plane[i] >= 0) p->normal_plane[i] = ctx->plane_orig[p->normal_plane[i]];
    }
    return LIVO2_OK;
  }
  if (total > ctx->h_pts_cap) {
    if (ctx->h_pts) HIPCHK(hipHostFree(ctx->h_pts));
    ctx->h_pts = nullptr; ctx->h_pts_cap = 0;
    HIPCHK(hipHostMalloc(&ctx->h_pts, total + total / 4));
    ctx->h_pts_cap = total + total / 4;
  }
  char *base = static_cast<char *>(ctx->h_pts);
  size_t off = 0, offs[NITEMS] = {};
  for (int k = 0; k < NITEMS; k++) {
    if (!items[k].dst) continue;
    offs[k] = off;
    if (items[k].bytes) HIPCHK(devalloc::memcpy_async(base + off, items[k].src, items[k].bytes, hipMemcpyDeviceToHost, ctx->stream));
    off += (items[k].bytes + 63) & ~(size_t)63;
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < NITEMS; k++) if (items[k].dst && items[k].bytes) std::memcpy(items[k].dst, base + offs[k], items[k].bytes);
  // device plane numbering (Morton order) -> the caller's
  if (!ctx->tree_mode) {
    if (p->match_plane) for (size_t i = 0; i < n; i++) if (p->match_plane[i] >= 0) p->match_plane[i] = ctx->plane_orig[p->match_plane[i]];
    if (p->normal_plane) for (size_t i = 0; i < n; i++) if (p->normal_plane[i] >= 0) p->normal_plane[i] = ctx->plane_orig[p->normal_plane[i]];
  }
  return LIVO2_OK;
}

// ---- co-residency of persistent grids -----------------------------------------------------------------------------------------------
// A grid that synchronises through an arrival counter must be resident as a whole.  Capacity = compute units x blocks per CU (occupancy query of the kernel);
// all contexts of this process on one device share it: a launch is admitted only if the blocks of the persistent launches still in flight (their completion
// events have not fired) plus its own fit, otherwise the caller takes the launch-per-step path.  Kernels that do not spin (everything else in this library) can
// only delay a persistent grid, never deadlock it.
struct PersistSlot { const livo2_ctx *owner; hipEvent_t ev; int blocks; int device; bool pending; };
std::mutex g_persist_mu;
std::vector<PersistSlot> g_persist;
int persist_capacity(int device) {
  static int cap[64] = {};
  if (device < 0 || device >= 64) return 0;
  if (cap[device] == 0) {
    hipDeviceProp_t prop; int per_cu = 0;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return 0;
    int per_cu_inv = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_visual_update_persistent<false>, VP_BLOCK, 0) != hipSuccess || per_cu < 1) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_inv, k_visual_update_persistent<true>, VP_BLOCK, 0) != hipSuccess || per_cu_inv < 1) return 0;
    per_cu = std::min(per_cu, per_cu_inv);
    cap[device] = prop.multiProcessorCount * std::min(per_cu, 2);        // a C4 frame takes 250 blocks (one per CU); a second grid of another context may share the CUs
  }
  return cap[device];
}
// Returns the grid size to launch (0: not admitted).  Admission RESERVES the blocks under the lock (slot marked pending until persist_register has recorded the
// event behind the launch): two host threads admitting at the same time must not both see a free device — two half-resident grids would wait for each other's
// words until the watchdog fires.
int persist_admit(livo2_ctx *ctx, int rows, int *halves) {
  const int cap = persist_capacity(ctx->device);
  if (cap <= 0) return 0;
  if (!ctx->vp_done && hipEventCreateWithFlags(&ctx->vp_done, hipEventDisableTiming) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lk(g_persist_mu);
  int busy = 0, own = -1, others = 0;
  for (size_t i = 0; i < g_persist.size();) {
    if (g_persist[i].owner == ctx) { own = (int)i; i++; continue; }            // this context's own earlier grid runs BEFORE the new one (same stream)
    if (g_persist[i].device == ctx->device) others++;
    if (!g_persist[i].pending && g_persist[i].blocks > 0 && hipEventQuery(g_persist[i].ev) == hipSuccess) g_persist[i].blocks = 0;      // done: the slot stays (its owner is alive)
    if (g_persist[i].device == ctx->device) busy += g_persist[i].blocks;
    i++;
  }
  // Block shape (the published rows, and so the results, are the same for both): alone on the device, one row per block (16 patches, the residual on one wave
  // per SIMD, 250 blocks for a C4 frame); as soon as another context of this process uses resident grids too, two rows per block (all 8 waves evaluate, 125
  // blocks) so that two updates fit on the device side by side instead of one of them falling back to the launch-per-step sequence.
  *halves = others > 0 ? 2 : 1;
  const int grid = (std::min(rows, VP_MAX_ROWS) + *halves - 1) / *halves;
  if (busy + grid > cap) return 0;
  if (own >= 0) { g_persist[own].blocks = std::max(g_persist[own].blocks, grid); g_persist[own].pending = true; }
  else g_persist.push_back(PersistSlot{ctx, ctx->vp_done, grid, ctx->device, true});
  return grid;
}
// after the launch (or instead of it, blocks = 0): the event now stands for the newest grid of this context.  The slot of a context lives until the context is
// destroyed (persist_forget): `others` above counts the contexts that use resident grids, not the grids in flight at one instant.
void persist_register(livo2_ctx *ctx, int blocks) {
  const bool recorded = blocks > 0 && hipEventRecord(ctx->vp_done, ctx->stream) == hipSuccess;
  std::lock_guard<std::mutex> lk(g_persist_mu);
  for (size_t i = 0; i < g_persist.size(); i++)
    if (g_persist[i].owner == ctx) { g_persist[i].blocks = recorded ? blocks : 0; g_persist[i].pending = false; return; }
}
void persist_forget(livo2_ctx *ctx) {
  std::lock_guard<std::mutex> lk(g_persist_mu);
  for (size_t i = 0; i < g_persist.size(); i++) if (g_persist[i].owner == ctx) { g_persist[i] = g_persist.back(); g_persist.pop_back(); break; }
}

int check_visual_cfg(livo2_ctx *ctx, const livo2_visual_cfg *cfg) {
  if (!cfg) return fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL");
  if (cfg->inverse_composition_en && ctx->M > 0 && !ctx->has_ref) return fail(ctx, LIVO2_ERR_INVALID, "inverse_composition_en needs livo2_visual_set_reference after set_frame");
  if (cfg->max_iterations < 1 || cfg->max_iterations > LIVO2_MAX_ITERS) return fail(ctx, LIVO2_ERR_INVALID, "max_iterations out of range");
  if (cfg->patch_pyrimid_level < 1 || cfg->patch_pyrimid_level > LIVO2_MAX_LEVELS) return fail(ctx, LIVO2_ERR_INVALID, "patch_pyrimid_level out of range");
  if (ctx->M > 0 && cfg->patch_pyrimid_level > ctx->L) return fail(ctx, LIVO2_ERR_INVALID, "patch_pyrimid_level exceeds the uploaded warp_patch levels");
  if (!(cfg->img_point_cov > 0)) return fail(ctx, LIVO2_ERR_INVALID, "img_point_cov must be > 0");
  if (cfg->mp_proc_num < 0 || cfg->mp_proc_num > LIVO2_WAVE) return fail(ctx, LIVO2_ERR_INVALID, "mp_proc_num out of [0,64]");
  if (cfg->cam.distortion < 0 || cfg->cam.distortion > LIVO2_CAM_EQUIDISTANT) return fail(ctx, LIVO2_ERR_INVALID, "cam.distortion must be 0 (pinhole), 1 (radtan) or 2 (equidistant)");
  return LIVO2_OK;
}

void m3mul(const double *A, const double *B, double *C) { for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j]; }

// initializeVIO constants (reference src/vio.cpp:27-38, 57-65)
VisualKernelArgs make_visual_args(livo2_ctx *ctx, const livo2_visual_cfg *cfg, int level) {
  VisualKernelArgs a{};
  a.img = ctx->d_img; a.width = ctx->width; a.height = ctx->height; a.stride = ctx->stride;
  a.pos = ctx->d_pos; a.warp = ctx->d_warp; a.search_levels = ctx->d_search; a.inv_expo = ctx->d_invexpo;
  a.M = ctx->M; a.L = ctx->L; a.level = level; a.exposure_en = cfg->exposure_estimate_en ? 1 : 0;
  a.fx = cfg->cam.fx; a.fy = cfg->cam.fy; a.cx = cfg->cam.cx; a.cy = cfg->cam.cy; std::memcpy(a.d, cfg->cam.d, 40); a.distortion = cfg->cam.distortion;
  double Rli[9], Pli[3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rli[i * 3 + j] = cfg->extR[j * 3 + i];                     // Rli = rot^T
  for (int i = 0; i < 3; i++) Pli[i] = ((-Rli[i * 3]) * cfg->extT[0] + (-Rli[i * 3 + 1]) * cfg->extT[1]) + (-Rli[i * 3 + 2]) * cfg->extT[2];   // Pli = -rot^T * transl
  m3mul(cfg->Rcl, Rli, a.Rci);                                                                                         // Rci = Rcl * Rli
  for (int i = 0; i < 3; i++) a.Pci[i] = ((cfg->Rcl[i * 3] * Pli[0] + cfg->Rcl[i * 3 + 1] * Pli[1]) + cfg->Rcl[i * 3 + 2] * Pli[2]) + cfg->Pcl[i];
  double Pic[3];
  for (int i = 0; i < 3; i++) Pic[i] = ((-a.Rci[i]) * a.Pci[0] + (-a.Rci[3 + i]) * a.Pci[1]) + (-a.Rci[6 + i]) * a.Pci[2];   // Pic = -Rci^T * Pci
  double tmp[9] = {0.0, -Pic[2], Pic[1], Pic[2], 0.0, -Pic[0], -Pic[1], Pic[0], 0.0}, nR[9];
  for (int i = 0; i < 9; i++) nR[i] = -a.Rci[i];
  m3mul(nR, tmp, a.Jdp_dR);                                                                                            // Jdp_dR = -Rci * skew(Pic)
  return a;
}

VisualRefArgs make_ref_args(livo2_ctx *ctx) {
  VisualRefArgs r{};
  r.ref_imgs = ctx->d_ref_imgs; r.ref_idx = ctx->d_ref_idx; r.ref_px = ctx->d_ref_px; r.ref_f = ctx->d_ref_f; r.ref_R = ctx->d_ref_R; r.ref_pos = ctx->d_ref_pos;
  r.gref = ctx->d_gref; r.mref = ctx->d_mref; r.n_ref = ctx->n_ref;
  return r;
}

int visual_grid(int M) { return std::max(1, (M + VIS_PPB - 1) / VIS_PPB); }          // forward-compositional kernels: VIS_PPB patches per block
int visual_grid_inverse(int M) { return std::max(1, (M + VIS_WAVES - 1) / VIS_WAVES); }   // inverse-compositional kernels: one patch per wave
// updateStateInverse has no OpenMP loop (vio.cpp:1422-1477): its frame error is always the serial sum
VisualSolveArgs visual_solve_args(livo2_ctx *ctx, const livo2_visual_cfg *cfg) {
  const int T = std::max(1, cfg->inverse_composition_en ? 1 : cfg->mp_proc_num);            // (updateStateInverse has no OpenMP loop: one serial chain, vio.cpp:1418-1477)
  return VisualSolveArgs{ctx->d_errors, ctx->M, ctx->visual_error_waves ? T : -T};
}

} // namespace

extern "C" {

const char *livo2_version(void) { return "livo2_hip 0.2 (gfx950)"; }     // 0.2: livo2_select_cfg.raycast_en (was pad), livo2_lidar_points.pinned (appended): bindings check livo2_abi_sizeof
int32_t livo2_abi_sizeof(const char *name) {
  if (!name) return 0;
#define LIVO2_SZ(T) if (std::strcmp(name, #T) == 0) return (int32_t)sizeof(T);
  LIVO2_SZ(livo2_state) LIVO2_SZ(livo2_map_view) LIVO2_SZ(livo2_lidar_cfg) LIVO2_SZ(livo2_lidar_sums) LIVO2_SZ(livo2_lidar_points) LIVO2_SZ(livo2_lidar_result)
  LIVO2_SZ(livo2_cam) LIVO2_SZ(livo2_visual_cfg) LIVO2_SZ(livo2_visual_sums) LIVO2_SZ(livo2_visual_step) LIVO2_SZ(livo2_visual_result)
  LIVO2_SZ(livo2_plane_fit) LIVO2_SZ(livo2_imu_step) LIVO2_SZ(livo2_imu_cfg) LIVO2_SZ(livo2_imu_pose) LIVO2_SZ(livo2_select_cfg)
  LIVO2_SZ(livo2_map_tree_cfg) LIVO2_SZ(livo2_retrieve_cfg) LIVO2_SZ(livo2_retrieve_candidates) LIVO2_SZ(livo2_retrieve_out) LIVO2_SZ(livo2_visual_obs) LIVO2_SZ(livo2_retrieve_chain_out) LIVO2_SZ(livo2_visual_map_delta) LIVO2_SZ(livo2_frame_in) LIVO2_SZ(livo2_visual_reference)
#undef LIVO2_SZ
  return 0;
}

static int ctx_create_impl(int device, void *stream, bool external, livo2_ctx **out) {
  if (!out) return LIVO2_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return LIVO2_ERR_NO_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LIVO2_ERR_NO_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return LIVO2_ERR_NO_DEVICE;      // kernels are built for gfx950 only
  if (hipSetDevice(device) != hipSuccess) return LIVO2_ERR_NO_DEVICE;
  livo2_ctx *ctx = new livo2_ctx;
  ctx->device = device;
  if (external) { ctx->stream = (hipStream_t)stream; ctx->own_stream = false; }
  else { if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return LIVO2_ERR_HIP; } ctx->own_stream = true; }
  if (DMALLOC((void **)&ctx->d_ctl, sizeof(DevCtl)) != hipSuccess || hipHostMalloc((void **)&ctx->h_in, sizeof(HostIn) * IN_RING) != hipSuccess ||
      hipHostMalloc(&ctx->h_out, sizeof(DevCtl)) != hipSuccess) { livo2_ctx_destroy(ctx); return LIVO2_ERR_HIP; }
  if (hipEventCreate(&ctx->span0) != hipSuccess || hipEventCreate(&ctx->span1) != hipSuccess) { livo2_ctx_destroy(ctx); return LIVO2_ERR_HIP; }
  for (int k = 0; k < IN_RING; k++) if (hipEventCreateWithFlags(&ctx->in_ev[k], hipEventDisableTiming) != hipSuccess) { livo2_ctx_destroy(ctx); return LIVO2_ERR_HIP; }
  if (hipMemsetAsync(ctx->d_ctl, 0, sizeof(DevCtl), ctx->stream) != hipSuccess) { livo2_ctx_destroy(ctx); return LIVO2_ERR_HIP; }
  *out = ctx;
  return LIVO2_OK;
}
int livo2_ctx_create(int device, livo2_ctx **out) { return ctx_create_impl(device, nullptr, false, out); }
int livo2_ctx_create_on_stream(int device, void *hip_stream, livo2_ctx **out) { return ctx_create_impl(device, hip_stream, true, out); }

void livo2_ctx_destroy(livo2_ctx *ctx) {
  if (!ctx) return;
  hipError_t e = hipSetDevice(ctx->device);
  if (ctx->stream) e = hipStreamSynchronize(ctx->stream);
  void *dev[] = {ctx->d_ctl, ctx->d_slots, ctx->d_cand, ctx->d_planes, ctx->d_planes_hot, ctx->d_plane_aux, ctx->d_cand_aux, ctx->d_xyz_aos, ctx->d_x, ctx->d_y, ctx->d_z,
                 ctx->d_cb, ctx->d_keys, ctx->d_keys2, ctx->d_idx, ctx->d_perm, ctx->d_sort_tmp, ctx->d_partials, ctx->d_match, ctx->d_normal_plane, ctx->d_dis, ctx->d_pw, ctx->d_var, ctx->d_rinv, ctx->d_hrow, ctx->d_img,
                 ctx->d_pos, ctx->d_invexpo, ctx->d_warp, ctx->d_search, ctx->d_errors, ctx->d_zdbg, ctx->d_Hdbg, ctx->d_ref_imgs, ctx->d_ref_idx, ctx->d_ref_px, ctx->d_ref_f, ctx->d_ref_R, ctx->d_ref_pos,
                 ctx->d_gref, ctx->d_mref, ctx->bd_xyz_aos, ctx->bd_x, ctx->bd_y, ctx->bd_z, ctx->bd_cb, ctx->bd_keys, ctx->bd_keys2, ctx->bd_idx, ctx->bd_perm, ctx->bd_partials,
                 ctx->bd_block_frame, ctx->bd_ctl, ctx->bd_entries, ctx->bd_in, ctx->bd_results, ctx->d_plane_internal, ctx->d_plane_cand_pos,
                 ctx->d_fit_pw, ctx->d_fit_var, ctx->d_fit_off, ctx->d_fit_idx, ctx->d_fit_out, ctx->d_fit_list,
                 ctx->d_c_pos, ctx->d_c_normal, ctx->d_c_px, ctx->d_c_f, ctx->d_c_R, ctx->d_c_t, ctx->d_c_ie, ctx->d_c_ncc, ctx->d_c_A, ctx->d_c_idx, ctx->d_c_lvl, ctx->d_c_acc,
                 ctx->d_c_sl, ctx->d_c_slot, ctx->d_c_count, ctx->d_c_err, ctx->d_c_patch, ctx->d_raw, ctx->d_curv, ctx->d_poses, ctx->d_vg_head, ctx->d_vg_slot, ctx->d_vg_misc,
                 ctx->d_vm_pos, ctx->d_vm_pkey, ctx->d_vm_active, ctx->d_vm_fov, ctx->d_sel_pg, ctx->d_sel_set, ctx->d_sel_depth, ctx->d_sel_best, ctx->d_sel_type, ctx->d_sel_point,
                 ctx->d_sel_flag, ctx->d_sel_dist, ctx->d_sel_disc, ctx->d_imu_steps, ctx->d_imu_poses, ctx->d_imu_state,
                 ctx->d_ob_off, ctx->d_ob_id, ctx->d_ob_img, ctx->d_ob_lvl, ctx->d_vm_refpatch, ctx->d_ob_px, ctx->d_ob_f, ctx->d_ob_R, ctx->d_ob_t, ctx->d_ob_ie, ctx->d_vm_normal,
                 ctx->d_ob_patch, ctx->d_vm_ninit, ctx->d_ob_imgs, ctx->d_ch_obs, ctx->d_ch_flag, ctx->d_ch_slot, ctx->d_cand_cell, ctx->d_cand_point, ctx->d_cand_obs,
                 ctx->d_sub_point, ctx->d_sub_obs, ctx->d_ch_count, ctx->d_c_id, ctx->d_c_leader, ctx->d_ld_keys, ctx->d_ld_vals,
                 ctx->vbd_img, ctx->vbd_pos, ctx->vbd_invexpo, ctx->vbd_partials, ctx->vbd_warp, ctx->vbd_errors, ctx->vbd_search, ctx->vbd_block_frame, ctx->vbd_entries, ctx->vbd_results,
                 ctx->mt_in_pw, ctx->mt_in_var, ctx->mt_keys, ctx->mt_keys2, ctx->mt_idx, ctx->mt_order, ctx->mt_head, ctx->mt_slot, ctx->mt_seg_begin, ctx->mt_seg_root, ctx->mt_nseg, ctx->mt_state,
                 ctx->mt.nodes, ctx->mt.pool_pw, ctx->mt.pool_var, ctx->mt.counters, ctx->mt.dirty_list, ctx->mt.overflow_list, ctx->d_vp_rows, ctx->d_vp_errs, ctx->d_vp_prof, ctx->mt_rp_rows, ctx->mt_rp_out, ctx->d_lpt_order, ctx->d_lpt_cost, ctx->d_lidar_tickets, ctx->d_ob_list, ctx->d_ob_cnt, ctx->d_delta, ctx->d_bcov_rows, ctx->d_vm_set, ctx->d_ray_set, ctx->d_ray_key, ctx->d_ray_hit_key, ctx->d_ray_hit_best, ctx->d_ray_action, ctx->d_ray_hit_cell,
                 ctx->d_ray_counters, ctx->d_ray_add, ctx->d_frame_arena, ctx->d_lidar_stamps, ctx->d_lidar_span_acc, ctx->mt_sort_tmp, ctx->d_ret_blob,
                 ctx->vbd_ref_imgs, ctx->vbd_ref_idx, ctx->vbd_ref_px, ctx->vbd_ref_f, ctx->vbd_ref_R, ctx->vbd_ref_pos, ctx->vbd_gref, ctx->vbd_mref};
  for (void *p : dev) if (p) e = DFREE(p);
  if (ctx->h_in) e = hipHostFree(ctx->h_in);
  if (ctx->h_out) e = hipHostFree(ctx->h_out);
  if (ctx->h_pts) e = hipHostFree(ctx->h_pts);
  if (ctx->h_ret) e = hipHostFree(ctx->h_ret);
  if (ctx->h_img) e = hipHostFree(ctx->h_img);
  if (ctx->h_err) e = hipHostFree(ctx->h_err);
  if (ctx->h_rp) e = hipHostFree(ctx->h_rp);
  if (ctx->img_ready) e = hipEventDestroy(ctx->img_ready);
  if (ctx->stream_img) e = hipStreamDestroy(ctx->stream_img);
  if (ctx->h_delta) e = hipHostFree(ctx->h_delta);
  if (ctx->h_frame_res) e = hipHostFree(ctx->h_frame_res);
  for (int k = 0; k < 2; k++) { if (ctx->frame_stage[k]) e = hipHostFree(ctx->frame_stage[k]); if (ctx->frame_stage_ev[k]) e = hipEventDestroy(ctx->frame_stage_ev[k]); if (ctx->frame_res_ev[k]) e = hipEventDestroy(ctx->frame_res_ev[k]); }
  if (ctx->delta_ev) e = hipEventDestroy(ctx->delta_ev);
  for (int k = 0; k < 2; k++) { if (ctx->scan_stage[k]) e = hipHostFree(ctx->scan_stage[k]); if (ctx->scan_stage_ev[k]) e = hipEventDestroy(ctx->scan_stage_ev[k]); }
  if (ctx->bh_in) e = hipHostFree(ctx->bh_in);
  if (ctx->bh_results) e = hipHostFree(ctx->bh_results);
  if (ctx->bh_entries) e = hipHostFree(ctx->bh_entries);
  if (ctx->vbh_entries) e = hipHostFree(ctx->vbh_entries);
  if (ctx->vbh_results) e = hipHostFree(ctx->vbh_results);
  for (auto &b : ctx->bins) for (auto &ev : b.used) { e = hipEventDestroy(ev.a); e = hipEventDestroy(ev.b); }
  for (auto &ev : ctx->ev_pool) { e = hipEventDestroy(ev.a); e = hipEventDestroy(ev.b); }
  for (int k = 0; k < IN_RING; k++) if (ctx->in_ev[k]) e = hipEventDestroy(ctx->in_ev[k]);
  if (ctx->vp_done) { persist_forget(ctx); e = hipEventDestroy(ctx->vp_done); }
  if (ctx->mt_worker.joinable()) { { std::lock_guard<std::mutex> lk(ctx->mt_mu); ctx->mt_quit = true; } ctx->mt_cv.notify_all(); ctx->mt_worker.join(); }
  if (ctx->h_mt_cnt) e = hipHostFree(ctx->h_mt_cnt);
  if (ctx->stream_mt) { e = hipStreamSynchronize(ctx->stream_mt); e = hipStreamDestroy(ctx->stream_mt); e = hipEventDestroy(ctx->mt_fork); e = hipEventDestroy(ctx->mt_ev0); e = hipEventDestroy(ctx->mt_ev1); }
  if (ctx->span0) e = hipEventDestroy(ctx->span0);
  if (ctx->span1) e = hipEventDestroy(ctx->span1);
  if (ctx->own_stream && ctx->stream) e = hipStreamDestroy(ctx->stream);
  (void)e;
  delete ctx;
}

const char *livo2_last_error(const livo2_ctx *ctx) { return ctx ? ctx->err.c_str() : "ctx is NULL"; }
void *livo2_ctx_stream(livo2_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int livo2_host_alloc_pinned(size_t bytes, void **out) {
  if (!out) return LIVO2_ERR_INVALID;
  *out = nullptr;
  return hipHostMalloc(out, bytes ? bytes : 1) == hipSuccess ? LIVO2_OK : LIVO2_ERR_HIP;
}
void livo2_host_free_pinned(void *p) { if (p) { hipError_t e = hipHostFree(p); (void)e; } }

int livo2_ctx_synchronize(livo2_ctx *ctx) { if (!ctx) return LIVO2_ERR_INVALID; HIPCHK(hipStreamSynchronize(ctx->stream)); return rz_gate(ctx); }

// checker self-test: one 4-byte store at `byte_offset` relative to the END of the ctx's control block (negative: relative to its start)
__global__ void k_rz_poke(char *p) { *reinterpret_cast<volatile uint32_t *>(p) = 0x600DF00Du; }
int livo2_debug_redzone_poke(livo2_ctx *ctx, int64_t byte_offset) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (devalloc::mode() == 0) return fail(ctx, LIVO2_ERR_INVALID, "LIVO2_REDZONE is not set: an out-of-bounds store would hit live memory");
  HIPCHK(hipSetDevice(ctx->device));
  char *base = reinterpret_cast<char *>(ctx->d_ctl);
  k_rz_poke<<<1, 1, 0, ctx->stream>>>(byte_offset >= 0 ? base + sizeof(DevCtl) + byte_offset : base + byte_offset);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LIVO2_OK;
}

int livo2_debug_redzone_check(livo2_ctx *ctx, int32_t *mode, int64_t *damaged_words) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (mode) *mode = devalloc::mode();
  if (damaged_words) *damaged_words = 0;
  if (devalloc::mode() != 1) return LIVO2_OK;
  HIPCHK(hipSetDevice(ctx->device));
  char msg[320];
  const long long bad = devalloc::check(msg, sizeof(msg));
  if (damaged_words) *damaged_words = bad;
  if (bad) { ctx->err = bad < 0 ? "redzone check could not run" : msg; return LIVO2_ERR_HIP; }
  return LIVO2_OK;
}

int livo2_ctx_set_option(livo2_ctx *ctx, const char *name, int32_t value) {
  if (!ctx || !name) return LIVO2_ERR_INVALID;
  if (std::strcmp(name, "lidar_block_order") == 0) { ctx->lidar_block_order = value != 0; ctx->lpt_valid = false; return LIVO2_OK; }
  if (std::strcmp(name, "lidar_fused_iteration") == 0) { ctx->lidar_fused = value != 0; ctx->lpt_valid = false; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent") == 0) { ctx->visual_persistent = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_inverse") == 0) { ctx->visual_persistent_inverse = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "visual_error_waves") == 0) { ctx->visual_error_waves = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "map_update_spread") == 0) {
    if (value != 0 && value != 1 && value != 2 && value != 4 && value != 8) return fail(ctx, LIVO2_ERR_INVALID, "map_update_spread: 0 (automatic), 1, 2, 4 or 8");
    ctx->mt_spread_opt = (int)value; return LIVO2_OK;
  }
  if (std::strcmp(name, "map_update_wide_fit") == 0) { ctx->mt_wide_fit_opt = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_timeout_us") == 0) {
    if (value < 100 || value > 10000000) return fail(ctx, LIVO2_ERR_INVALID, "visual_persistent_timeout_us out of [100, 10000000]");
    ctx->vp_timeout_us = value; return LIVO2_OK;
  }
  if (std::strcmp(name, "scan_small_fused") == 0) { ctx->scan_small_fused = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "frame_ingest") == 0) {
    if (value < 0 || value > 2) return fail(ctx, LIVO2_ERR_INVALID, "frame_ingest must be 0 (a copy per array), 1 (arena + scatter launch) or 2 (launch reads the pinned block)");
    ctx->frame_ingest = value; return LIVO2_OK;
  }
  if (std::strcmp(name, "frame_publish") == 0) { ctx->frame_publish = value != 0; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_debug_timeout") == 0) { ctx->vp_debug_timeout = value != 0; return LIVO2_OK; }     // test hook: the last block of the grid leaves at once, the others give up after 2 ms
  return fail(ctx, LIVO2_ERR_INVALID, "unknown option");
}
int livo2_ctx_get_counter(livo2_ctx *ctx, const char *name, int64_t *value) {
  if (!ctx || !name || !value) return LIVO2_ERR_INVALID;
  if (std::strcmp(name, "visual_persistent_launches") == 0) { *value = ctx->vp_used; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_fallbacks") == 0) { *value = ctx->vp_fallback; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_timeouts") == 0) { *value = ctx->vp_timeouts; return LIVO2_OK; }
  if (std::strcmp(name, "visual_persistent_backoff_skips") == 0) { *value = ctx->vp_backoff_skips; return LIVO2_OK; }
  if (std::strcmp(name, "map_tree_grow_events") == 0) { *value = ctx->mt_grow_events; return LIVO2_OK; }
  if (std::strcmp(name, "lidar_fused_launches") == 0) { *value = ctx->lidar_fused_launches; return LIVO2_OK; }
  if (std::strcmp(name, "visual_map_delta_calls") == 0) { *value = ctx->vm_delta_calls; return LIVO2_OK; }
  if (std::strcmp(name, "scan_small_launches") == 0) { *value = ctx->scan_small_launches; return LIVO2_OK; }
  if (std::strcmp(name, "frame_ingest_launches") == 0) { *value = ctx->frame_ingest_launches; return LIVO2_OK; }
  if (std::strcmp(name, "frame_zero_copy_launches") == 0) { *value = ctx->frame_zero_copy_launches; return LIVO2_OK; }
  if (std::strcmp(name, "frame_publish_launches") == 0) { *value = ctx->frame_publish_launches; return LIVO2_OK; }
  if (std::strcmp(name, "visual_map_delta_grows") == 0) { *value = ctx->vm_delta_grows; return LIVO2_OK; }
  {   // k_lidar_residual's own duration on the device clock, summed over the executed launches of the timed passes so far (10-ns ticks / launches / longest / shortest)
    static const char *const names[4] = {"lidar_residual_device_ticks", "lidar_residual_device_launches", "lidar_residual_device_ticks_max", "lidar_residual_device_ticks_min"};
    for (int k = 0; k < 4; k++) if (std::strcmp(name, names[k]) == 0) {
      *value = 0;
      if (!ctx->d_lidar_span_acc) return LIVO2_OK;
      unsigned long long h[4];
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(h, ctx->d_lidar_span_acc, 32, hipMemcpyDeviceToHost));
      *value = (int64_t)h[k];
      return LIVO2_OK;
    }
  }
  return fail(ctx, LIVO2_ERR_INVALID, "unknown counter");
}

int livo2_ctx_kernel_timing(livo2_ctx *ctx, int enable) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (enable) {      // the stamps of k_lidar_residual (its own duration on the device clock): room for the current scan, emptied
    HIPCHK(hipSetDevice(ctx->device));
    const size_t need = 2 * (size_t)std::max(lidar_grid(std::max(ctx->n, 1), ctx->lidar_block), 64);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    int rc = ensure(ctx, ctx->d_lidar_stamps, ctx->lidar_stamps_cap, need); if (rc) return rc;
    if (!ctx->d_lidar_span_acc) { HIPCHK(DMALLOC((void **)&ctx->d_lidar_span_acc, 32)); HIPCHK(hipMemsetAsync(ctx->d_lidar_span_acc, 0, 32, ctx->stream)); }
    HIPCHK(hipMemsetAsync(ctx->d_lidar_stamps, 0xFF, ctx->lidar_stamps_cap * 8, ctx->stream));
  }
  ctx->timing = enable != 0;
  return LIVO2_OK;
}
int livo2_ctx_kernel_timing_read(livo2_ctx *ctx, int which, double *total_ms, int64_t *launches, int reset) {
  if (!ctx || which < 0 || which > 3) return LIVO2_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  TimingBin &b = ctx->bins[which];
  for (auto &ev : b.used) { float ms = 0; if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) { b.total_ms += ms; b.launches++; } ctx->ev_pool.push_back(ev); }
  b.used.clear();
  if (total_ms) *total_ms = b.total_ms;
  if (launches) *launches = b.launches;
  if (reset) { b.total_ms = 0; b.launches = 0; }
  return LIVO2_OK;
}

static int mt_join(livo2_ctx *ctx);      // api_map_tree.inc: wait for a map update that runs on the second stream (no-op when none is pending)
// ---- the entry points, by subsystem (order matters: later parts call helpers of earlier ones) ----------------------------------------------------------------
#include "api_map.inc"
#include "api_imu.inc"
#include "api_map_tree.inc"
#include "api_lidar.inc"
#include "api_retrieve.inc"
#include "api_visual.inc"

} // extern "C"
