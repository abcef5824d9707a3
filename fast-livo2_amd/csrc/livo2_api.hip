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
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
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
  livo2_visual_cfg frame_vcfg[2] = {}; int frame_M[2] = {0, 0}; bool frame_persistent[2] = {false, false};
  bool vp_last_chained = false;
  int32_t *d_ch_obs = nullptr, *d_ch_flag = nullptr, *d_ch_slot = nullptr, *d_cand_cell = nullptr, *d_cand_point = nullptr, *d_cand_obs = nullptr, *d_sub_point = nullptr,
          *d_sub_obs = nullptr, *d_ch_count = nullptr;
  size_t ch_obs_cap = 0, ch_flag_cap = 0, ch_slot_cap = 0, cand_cell_cap = 0, cand_point_cap = 0, cand_obs_cap = 0, sub_point_cap = 0, sub_obs_cap = 0;
  int32_t *d_c_id = nullptr, *d_c_leader = nullptr, *d_ld_keys = nullptr, *d_ld_vals = nullptr; size_t c_id_cap = 0, c_leader_cap = 0, ld_keys_cap = 0, ld_vals_cap = 0;
  double chain_kernel_us = 0.0;
  // IMU propagation (N4)
  double *d_imu_steps = nullptr, *d_imu_poses = nullptr; size_t imu_steps_cap = 0, imu_poses_cap = 0; livo2_state *d_imu_state = nullptr;   // [2]: in, out
  double imu_kernel_us = 0.0;
  // scan
  bool has_scan = false;
  int n = 0, n_cap = 0, lidar_block = 256;
  float *d_xyz_aos = nullptr, *d_x = nullptr, *d_y = nullptr, *d_z = nullptr; double *d_cb = nullptr;
  uint32_t *d_keys = nullptr, *d_keys2 = nullptr; int32_t *d_idx = nullptr, *d_perm = nullptr; void *d_sort_tmp = nullptr; size_t sort_tmp_bytes = 0;
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
  bool has_vbatch = false;
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
  unsigned long long *d_vp_rows = nullptr; size_t vp_rows_cap = 0; unsigned long long *d_vp_errs = nullptr; size_t vp_errs_cap = 0; uint32_t vp_seq = 0;
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
  int mt_pv_n = -1;                         // points of the pv_list in mt_in_pw / mt_in_var (last map-tree update), -1: none since the last set_scan
  int32_t *mt_rp_rows = nullptr; size_t mt_rp_rows_cap = 0; double *mt_rp_out = nullptr; size_t mt_rp_out_cap = 0;   // livo2_map_tree_read_planes staging
  double mt_kernel_us = 0.0;
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

#define HIPCHK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t e_ = (call);                                                                                 \
    if (e_ != hipSuccess) {                                                                                 \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                                         \
      return LIVO2_ERR_HIP;                                                                                 \
    }                                                                                                       \
  } while (0)

int fail(livo2_ctx *ctx, int code, const char *msg) { if (ctx) ctx->err = msg; return code; }

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
#define ensure(...) ensure_at(__LINE__, __VA_ARGS__)

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
#define grow_array(...) grow_array_at(__LINE__, __VA_ARGS__)

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
#define keep_grow(...) keep_grow_at(__LINE__, __VA_ARGS__)
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
  if (resident > 0) {
    if (ctx->lidar_block == 128) hipLaunchKernelGGL(k_lidar_residual_resident<128>, dim3(grid), dim3(128), LIDAR_LDS_BYTES_OF(128) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
    else hipLaunchKernelGGL(k_lidar_residual_resident<256>, dim3(grid), dim3(256), LIDAR_LDS_BYTES_OF(256) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
  } else if (ctx->lidar_block == 128) hipLaunchKernelGGL(k_lidar_residual<128>, dim3(chunks), dim3(128), LIDAR_LDS_BYTES_OF(128) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
  else hipLaunchKernelGGL(k_lidar_residual<256>, dim3(chunks), dim3(256), LIDAR_LDS_BYTES_OF(256) + LIDAR_LDS_DUMP, ctx->stream, a, ctx->d_ctl, ctx->d_partials, check_stop, chunks, order, cost);
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

LidarKernelArgs make_lidar_args(livo2_ctx *ctx, const livo2_lidar_cfg *cfg) {
  LidarKernelArgs a{};
  a.x = ctx->d_x; a.y = ctx->d_y; a.z = ctx->d_z; a.cb = ctx->d_cb; a.perm = ctx->d_perm; a.n = ctx->n; a.max_layer = cfg->max_layer; a.map = ctx->map;
  a.voxel_size = cfg->voxel_size; a.sigma_num = cfg->sigma_num;
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
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_visual_update_persistent, VP_BLOCK, 0) != hipSuccess || per_cu < 1) return 0;
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
VisualSolveArgs visual_solve_args(livo2_ctx *ctx, const livo2_visual_cfg *cfg) { return VisualSolveArgs{ctx->d_errors, ctx->M, cfg->inverse_composition_en ? 1 : cfg->mp_proc_num}; }

} // namespace

extern "C" {

const char *livo2_version(void) { return "livo2_hip 0.2 (gfx950)"; }     // 0.2: livo2_select_cfg.raycast_en (was pad), livo2_lidar_points.pinned (appended): bindings check livo2_abi_sizeof
int32_t livo2_abi_sizeof(const char *name) {
  if (!name) return 0;
#define LIVO2_SZ(T) if (std::strcmp(name, #T) == 0) return (int32_t)sizeof(T);
  LIVO2_SZ(livo2_state) LIVO2_SZ(livo2_map_view) LIVO2_SZ(livo2_lidar_cfg) LIVO2_SZ(livo2_lidar_sums) LIVO2_SZ(livo2_lidar_points) LIVO2_SZ(livo2_lidar_result)
  LIVO2_SZ(livo2_cam) LIVO2_SZ(livo2_visual_cfg) LIVO2_SZ(livo2_visual_sums) LIVO2_SZ(livo2_visual_step) LIVO2_SZ(livo2_visual_result)
  LIVO2_SZ(livo2_plane_fit) LIVO2_SZ(livo2_imu_step) LIVO2_SZ(livo2_imu_cfg) LIVO2_SZ(livo2_imu_pose) LIVO2_SZ(livo2_select_cfg)
  LIVO2_SZ(livo2_map_tree_cfg) LIVO2_SZ(livo2_retrieve_cfg) LIVO2_SZ(livo2_retrieve_candidates) LIVO2_SZ(livo2_retrieve_out) LIVO2_SZ(livo2_visual_obs) LIVO2_SZ(livo2_retrieve_chain_out) LIVO2_SZ(livo2_visual_map_delta) LIVO2_SZ(livo2_frame_in)
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
                 ctx->d_ray_counters, ctx->d_ray_add};
  for (void *p : dev) if (p) e = DFREE(p);
  if (ctx->h_in) e = hipHostFree(ctx->h_in);
  if (ctx->h_out) e = hipHostFree(ctx->h_out);
  if (ctx->h_pts) e = hipHostFree(ctx->h_pts);
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
  if (std::strcmp(name, "visual_persistent_timeout_us") == 0) {
    if (value < 100 || value > 10000000) return fail(ctx, LIVO2_ERR_INVALID, "visual_persistent_timeout_us out of [100, 10000000]");
    ctx->vp_timeout_us = value; return LIVO2_OK;
  }
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
  if (std::strcmp(name, "visual_map_delta_grows") == 0) { *value = ctx->vm_delta_grows; return LIVO2_OK; }
  return fail(ctx, LIVO2_ERR_INVALID, "unknown counter");
}

int livo2_ctx_kernel_timing(livo2_ctx *ctx, int enable) { if (!ctx) return LIVO2_ERR_INVALID; ctx->timing = enable != 0; return LIVO2_OK; }
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

// ---- map ----------------------------------------------------------------------------------------------------------------
int livo2_map_upload(livo2_ctx *ctx, const livo2_map_view *m) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!m || m->n_roots < 0 || m->n_nodes < 0 || m->n_planes < 0) return fail(ctx, LIVO2_ERR_INVALID, "bad map view");
  if (m->n_roots > 0 && (!m->root_key || !m->root_node || !m->root_center || !m->root_quarter)) return fail(ctx, LIVO2_ERR_INVALID, "NULL root arrays");
  if (m->n_nodes > 0 && (!m->node_plane || !m->node_child)) return fail(ctx, LIVO2_ERR_INVALID, "NULL node arrays");
  if (m->n_planes > 0 && (!m->plane_normal || !m->plane_center || !m->plane_var || !m->plane_d || !m->plane_radius)) return fail(ctx, LIVO2_ERR_INVALID, "NULL plane arrays");
  HIPCHK(hipSetDevice(ctx->device));
  for (int i = 0; i < m->n_nodes; i++) {
    if (m->node_plane[i] < -1 || m->node_plane[i] >= m->n_planes) return fail(ctx, LIVO2_ERR_INVALID, "node_plane index out of range");
    for (int k = 0; k < 8; k++) { int c = m->node_child[(size_t)i * 8 + k]; if (c < -1 || c >= m->n_nodes) return fail(ctx, LIVO2_ERR_INVALID, "node_child index out of range"); }
  }
  if (m->n_planes >= (1 << CAND_LAYER_SHIFT)) return fail(ctx, LIVO2_ERR_RANGE, "too many planes");
  // Device plane order: roots along a Morton curve of their voxel key, each root's planes in depth-first order.  The scan is
  // Morton-sorted too, so neighbouring lanes gather neighbouring (often contiguous) records: measured on the C2 scene the
  // plane gather drops from ~random-access cost to within a few % of the fully contiguous floor.  The caller's plane numbering is
  // kept at the boundary through plane_internal / plane_orig.
  std::vector<int32_t> order(m->n_roots);
  {
    std::vector<std::pair<uint64_t, int32_t>> keyed(m->n_roots);
    auto spread21 = [](uint64_t v) { v &= 0x1fffffull; v = (v | v << 32) & 0x1f00000000ffffull; v = (v | v << 16) & 0x1f0000ff0000ffull; v = (v | v << 8) & 0x100f00f00f00f00full;
                                     v = (v | v << 4) & 0x10c30c30c30c30c3ull; v = (v | v << 2) & 0x1249249249249249ull; return v; };
    for (int r = 0; r < m->n_roots; r++) {
      const uint64_t x = (uint64_t)(m->root_key[(size_t)r * 3] + (1 << 20)), y = (uint64_t)(m->root_key[(size_t)r * 3 + 1] + (1 << 20)), z = (uint64_t)(m->root_key[(size_t)r * 3 + 2] + (1 << 20));
      keyed[r] = {spread21(x) | (spread21(y) << 1) | (spread21(z) << 2), r};
    }
    std::sort(keyed.begin(), keyed.end());
    for (int r = 0; r < m->n_roots; r++) order[r] = keyed[r].second;
  }
  std::vector<int32_t> pin((size_t)std::max(1, m->n_planes), -1);       // caller index -> device index
  int32_t next_plane = 0;
  // flatten every non-plane root into the depth-first list of its descendant planes (the walk of voxel_map.cpp:769-785)
  std::vector<int32_t> cand;
  std::vector<int32_t> cbegin(m->n_roots, 0), ccount(m->n_roots, 0);
  {
    struct Fr { int node, layer, next; };
    std::vector<Fr> st;
    for (int oi = 0; oi < m->n_roots; oi++) {
      const int r = order[oi];
      int root = m->root_node[r];
      if (root < 0 || root >= m->n_nodes) return fail(ctx, LIVO2_ERR_INVALID, "root_node index out of range");
      cbegin[r] = (int32_t)cand.size();
      if (m->node_plane[root] >= 0 && pin[m->node_plane[root]] < 0) pin[m->node_plane[root]] = next_plane++;
      if (m->node_plane[root] < 0) {
        st.clear(); st.push_back({root, 0, 0});
        while (!st.empty()) {
          Fr &f = st.back();
          if (f.next >= 8 || f.layer >= LIVO2_MAX_LAYER) { st.pop_back(); continue; }
          int child = m->node_child[(size_t)f.node * 8 + f.next++];
          if (child < 0) continue;
          int cl = f.layer + 1;
          int pl = m->node_plane[child];
          if (pl >= 0) { if (pin[pl] < 0) pin[pl] = next_plane++; cand.push_back(pin[pl] | (cl << CAND_LAYER_SHIFT)); }
          else { if (st.size() > 64) return fail(ctx, LIVO2_ERR_INVALID, "octree deeper than supported / cyclic"); st.push_back({child, cl, 0}); }
        }
      }
      ccount[r] = (int32_t)cand.size() - cbegin[r];
    }
    for (int p = 0; p < m->n_planes; p++) if (pin[p] < 0) pin[p] = next_plane++;      // planes no root reaches
  }
  // 2-choice cuckoo table, load factor <= 0.25, one 64-B slot per bucket
  uint32_t cap = 64;
  while (cap < (uint32_t)m->n_roots * 4u) cap <<= 1;
  std::vector<RootSlot> slots;
  uint32_t seed1 = 0x243f6a88u, seed2 = 0x85a308d3u;
  for (int attempt = 0;; attempt++) {
    if (attempt > 24) return fail(ctx, LIVO2_ERR_INVALID, "cuckoo placement failed");
    if (attempt > 0 && attempt % 6 == 0) cap <<= 1;
    slots.assign(cap, RootSlot{});
    for (auto &sl : slots) sl.val = -1;
    bool ok = true;
    for (int r = 0; r < m->n_roots && ok; r++) {
      int64_t kx = m->root_key[(size_t)r * 3], ky = m->root_key[(size_t)r * 3 + 1], kz = m->root_key[(size_t)r * 3 + 2];
      if (kx < INT32_MIN || kx > INT32_MAX || ky < INT32_MIN || ky > INT32_MAX || kz < INT32_MIN || kz > INT32_MAX) return fail(ctx, LIVO2_ERR_RANGE, "voxel key outside int32");
      RootSlot cur{};
      cur.kx = (int32_t)kx; cur.ky = (int32_t)ky; cur.kz = (int32_t)kz;
      int pl = m->node_plane[m->root_node[r]];
      cur.val = pl >= 0 ? pin[pl] : -2;
      for (int k = 0; k < 3; k++) cur.center[k] = m->root_center[(size_t)r * 3 + k];
      cur.quarter = m->root_quarter[r]; cur.cand_begin = cbegin[r]; cur.cand_count = ccount[r];
      {   // duplicate keys are a caller error
        uint32_t a = voxel_hash(cur.kx, cur.ky, cur.kz, seed1) & (cap - 1), b = voxel_hash(cur.kx, cur.ky, cur.kz, seed2) & (cap - 1);
        for (uint32_t h : {a, b}) if (slots[h].val != -1 && slots[h].kx == cur.kx && slots[h].ky == cur.ky && slots[h].kz == cur.kz) return fail(ctx, LIVO2_ERR_INVALID, "duplicate voxel key");
      }
      uint32_t h = voxel_hash(cur.kx, cur.ky, cur.kz, seed1) & (cap - 1);
      bool placed = false;
      for (int kick = 0; kick < 512; kick++) {
        if (slots[h].val == -1) { slots[h] = cur; placed = true; break; }
        std::swap(cur, slots[h]);                     // evict the resident, move it to its other bucket
        uint32_t a = voxel_hash(cur.kx, cur.ky, cur.kz, seed1) & (cap - 1), b = voxel_hash(cur.kx, cur.ky, cur.kz, seed2) & (cap - 1);
        h = (h == a) ? b : a;
      }
      if (!placed) ok = false;
    }
    if (ok) break;
    seed1 = seed1 * 1664525u + 1013904223u; seed2 = seed2 * 22695477u + 1u;
  }
  std::vector<double> recs((size_t)std::max(1, m->n_planes) * PLANE_REC_DOUBLES, 0.0);
  for (int p = 0; p < m->n_planes; p++)
    pack_plane(&recs[(size_t)pin[p] * PLANE_REC_DOUBLES], m->plane_normal + (size_t)p * 3, m->plane_center + (size_t)p * 3, m->plane_var + (size_t)p * 36, m->plane_d[p], m->plane_radius[p]);
  ctx->plane_internal = pin;
  ctx->plane_orig.assign(pin.size(), 0);
  for (int p = 0; p < m->n_planes; p++) ctx->plane_orig[pin[p]] = p;
  // candidate lists hold whole record copies (one round trip per evaluated pair); remember where each plane sits for
  // livo2_map_update_planes
  ctx->plane_cand_pos.assign((size_t)std::max(1, m->n_planes), -1);
  for (size_t k = 0; k < cand.size(); k++) ctx->plane_cand_pos[ctx->plane_orig[cand[k] & CAND_PLANE_MASK]] = (int32_t)k;

  HIPCHK(hipStreamSynchronize(ctx->stream));
  free_map_arrays(ctx);
  ctx->has_map = false;
  const size_t n_rows = (size_t)std::max(1, m->n_planes), n_cand = std::max<size_t>(1, cand.size());
  HIPCHK(DMALLOC((void **)&ctx->d_slots, (size_t)cap * sizeof(RootSlot)));
  HIPCHK(DMALLOC((void **)&ctx->d_cand, n_cand * PLANE_HOT_DOUBLES * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_cand_aux, n_cand * sizeof(PlaneAux)));
  HIPCHK(DMALLOC((void **)&ctx->d_planes, recs.size() * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_planes_hot, n_rows * PLANE_HOT_DOUBLES * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_plane_aux, n_rows * sizeof(PlaneAux)));
  HIPCHK(hipMemcpy(ctx->d_slots, slots.data(), (size_t)cap * sizeof(RootSlot), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(ctx->d_planes, recs.data(), recs.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemsetAsync(ctx->d_cand, 0, n_cand * PLANE_HOT_DOUBLES * 8, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_cand_aux, 0, n_cand * sizeof(PlaneAux), ctx->stream));
  // the residual kernel's view of the table is derived on the device from the master records
  hipLaunchKernelGGL(k_planes_hot, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_planes, (const int32_t *)nullptr, (const int32_t *)nullptr, (int)n_rows,
                     ctx->d_planes_hot, ctx->d_plane_aux, ctx->d_cand, ctx->d_cand_aux);
  if (!cand.empty()) {
    int32_t *d_meta = nullptr;
    HIPCHK(DMALLOC((void **)&d_meta, cand.size() * 4));
    HIPCHK(devalloc::memcpy_async(d_meta, cand.data(), cand.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_cand_fill, dim3((unsigned)((cand.size() * 8 + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_planes_hot, ctx->d_plane_aux, d_meta, (int)cand.size(), ctx->d_cand, ctx->d_cand_aux);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    HIPCHK(DFREE(d_meta));
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  set_map_view(ctx);
  ctx->map.mask = cap - 1;
  ctx->map.seed1 = seed1; ctx->map.seed2 = seed2; ctx->map.n_planes = m->n_planes;
  ctx->has_map = true;
  ctx->plane_tabs_fresh = false;
  if (ctx->tree_mode) {            // the snapshot replaces a device-resident tree
    hipError_t e2 = DFREE(ctx->mt.nodes); e2 = DFREE(ctx->mt.pool_pw); e2 = DFREE(ctx->mt.pool_var); e2 = DFREE(ctx->mt.counters); e2 = DFREE(ctx->mt.dirty_list); e2 = DFREE(ctx->mt.overflow_list); (void)e2;
    ctx->mt = MapTreeArgs{}; ctx->tree_mode = false;
  }
  return LIVO2_OK;
}

int livo2_map_update_planes(livo2_ctx *ctx, const int32_t *plane_idx, int32_t n, const double *normal, const double *center, const double *plane_var,
                            const float *d, const float *radius) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->has_map) return fail(ctx, LIVO2_ERR_NO_MAP, "no map uploaded");
  if (n < 0 || (n > 0 && (!plane_idx || !normal || !center || !plane_var || !d || !radius))) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (n == 0) return LIVO2_OK;
  for (int i = 0; i < n; i++) if (plane_idx[i] < 0 || plane_idx[i] >= ctx->map.n_planes) return fail(ctx, LIVO2_ERR_INVALID, "plane index out of range");
  HIPCHK(hipSetDevice(ctx->device));
  std::vector<double> recs((size_t)n * PLANE_REC_DOUBLES);
  for (int p = 0; p < n; p++) pack_plane(&recs[(size_t)p * PLANE_REC_DOUBLES], normal + (size_t)p * 3, center + (size_t)p * 3, plane_var + (size_t)p * 36, d[p], radius[p]);
  std::vector<int32_t> gpos(n), didx(n);
  for (int p = 0; p < n; p++) { gpos[p] = ctx->plane_cand_pos[plane_idx[p]]; didx[p] = ctx->plane_internal[plane_idx[p]]; }
  double *d_recs = nullptr; int32_t *d_idx = nullptr, *d_gpos = nullptr;
  HIPCHK(DMALLOC((void **)&d_recs, recs.size() * 8));
  HIPCHK(DMALLOC((void **)&d_idx, (size_t)n * 4));
  HIPCHK(DMALLOC((void **)&d_gpos, (size_t)n * 4));
  HIPCHK(devalloc::memcpy_async(d_recs, recs.data(), recs.size() * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(d_idx, didx.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(d_gpos, gpos.data(), (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_scatter_planes, dim3((n * 32 + 255) / 256), dim3(256), 0, ctx->stream, d_recs, d_idx, n, ctx->d_planes);
  hipLaunchKernelGGL(k_planes_hot, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_planes, d_idx, d_gpos, n, ctx->d_planes_hot, ctx->d_plane_aux, ctx->d_cand, ctx->d_cand_aux);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(DFREE(d_recs)); HIPCHK(DFREE(d_idx)); HIPCHK(DFREE(d_gpos));
  return LIVO2_OK;
}

// ---- IMU forward propagation -----------------------------------------------------------------------------------------------------
static ImuKernelArgs make_imu_args(livo2_ctx *ctx, const livo2_imu_cfg *cfg, int n, double *poses) {
  ImuKernelArgs a{};
  a.steps = ctx->d_imu_steps; a.n = n; a.ba_bg_est_en = cfg->ba_bg_est_en; a.gravity_est_en = cfg->gravity_est_en; a.exposure_estimate_en = cfg->exposure_estimate_en; a.first_call = cfg->first_call ? 1 : 0;
  std::memcpy(a.cov_gyr, cfg->cov_gyr, 24); std::memcpy(a.cov_acc, cfg->cov_acc, 24); std::memcpy(a.cov_bias_gyr, cfg->cov_bias_gyr, 24); std::memcpy(a.cov_bias_acc, cfg->cov_bias_acc, 24);
  a.cov_inv_expo = cfg->cov_inv_expo; a.G_m_s2 = cfg->G_m_s2; a.mean_acc_norm = cfg->mean_acc_norm;
  a.in = ctx->d_imu_state; a.out = ctx->d_imu_state + 1; a.poses = poses;
  return a;
}

int livo2_imu_propagate(livo2_ctx *ctx, const livo2_state *state_in, const livo2_imu_step *steps, int32_t n, const livo2_imu_cfg *cfg, livo2_state *state_out,
                        livo2_imu_pose *poses) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!state_in || !state_out || !cfg || n < 0 || n > 65536 || (n > 0 && (!steps || !poses))) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (!(cfg->mean_acc_norm > 0)) return fail(ctx, LIVO2_ERR_INVALID, "mean_acc_norm must be > 0");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc;
  if ((rc = ensure(ctx, ctx->d_imu_steps, ctx->imu_steps_cap, std::max((size_t)n * 8, (size_t)8)))) return rc;
  if ((rc = ensure(ctx, ctx->d_imu_poses, ctx->imu_poses_cap, std::max((size_t)n * 22, (size_t)22)))) return rc;
  if (!ctx->d_imu_state) HIPCHK(DMALLOC((void **)&ctx->d_imu_state, 2 * sizeof(livo2_state)));
  static_assert(sizeof(livo2_imu_step) == 64, "livo2_imu_step is 8 doubles");
  HIPCHK(devalloc::memcpy_async(ctx->d_imu_state, state_in, sizeof(livo2_state), hipMemcpyHostToDevice, ctx->stream));
  if (n > 0) HIPCHK(devalloc::memcpy_async(ctx->d_imu_steps, steps, (size_t)n * 64, hipMemcpyHostToDevice, ctx->stream));
  ImuKernelArgs a = make_imu_args(ctx, cfg, n, ctx->d_imu_poses);
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  hipLaunchKernelGGL(k_imu_propagate, dim3(1), dim3(IMU_THREADS), 0, ctx->stream, a);
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(devalloc::memcpy_async(state_out, ctx->d_imu_state + 1, sizeof(livo2_state), hipMemcpyDeviceToHost, ctx->stream));
  if (n > 0) HIPCHK(devalloc::memcpy_async(poses, ctx->d_imu_poses, (size_t)n * sizeof(livo2_imu_pose), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->imu_kernel_us = 1e3 * ms;
  return LIVO2_OK;
}
double livo2_imu_propagate_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->imu_kernel_us : 0.0; }

// ---- map maintenance: batched plane fit ----------------------------------------------------------------------------------------
int livo2_plane_fit_batch(livo2_ctx *ctx, const double *point_w, const double *var, const int32_t *offsets, int32_t n_groups, float planer_threshold,
                          const int32_t *plane_idx, livo2_plane_fit *out) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n_groups < 0 || (n_groups > 0 && (!offsets || !out))) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (n_groups == 0) return LIVO2_OK;
  if (offsets[0] != 0) return fail(ctx, LIVO2_ERR_INVALID, "offsets[0] must be 0");
  for (int g = 0; g < n_groups; g++) if (offsets[g + 1] < offsets[g]) return fail(ctx, LIVO2_ERR_INVALID, "offsets must be non-decreasing");
  const size_t N = (size_t)offsets[n_groups];
  if (N > 0 && (!point_w || !var)) return fail(ctx, LIVO2_ERR_INVALID, "point_w / var is NULL");
  if (plane_idx) {
    if (!ctx->has_map) return fail(ctx, LIVO2_ERR_NO_MAP, "plane_idx given but no map uploaded");
    for (int g = 0; g < n_groups; g++) if (plane_idx[g] >= ctx->map.n_planes) return fail(ctx, LIVO2_ERR_INVALID, "plane index out of range");
  }
  HIPCHK(hipSetDevice(ctx->device));
  int rc;
  if ((rc = ensure(ctx, ctx->d_fit_pw, ctx->fit_pw_cap, std::max(N * 3, (size_t)3)))) return rc;
  if ((rc = ensure(ctx, ctx->d_fit_var, ctx->fit_var_cap, std::max(N * 9, (size_t)9)))) return rc;
  if ((rc = ensure(ctx, ctx->d_fit_off, ctx->fit_off_cap, (size_t)n_groups + 1))) return rc;
  if ((rc = ensure(ctx, ctx->d_fit_idx, ctx->fit_idx_cap, (size_t)n_groups))) return rc;
  if ((rc = ensure(ctx, ctx->d_fit_out, ctx->fit_out_cap, (size_t)n_groups))) return rc;
  if ((rc = ensure(ctx, ctx->d_fit_list, ctx->fit_list_cap, (size_t)n_groups))) return rc;
  if (plane_idx && !ctx->plane_tabs_fresh) {
    const size_t np = ctx->plane_internal.size();
    if ((rc = ensure(ctx, ctx->d_plane_internal, ctx->plane_tab_cap, std::max(np, (size_t)1)))) return rc;
    if ((rc = ensure(ctx, ctx->d_plane_cand_pos, ctx->plane_tab_cap2, std::max(np, (size_t)1)))) return rc;
    HIPCHK(devalloc::memcpy_async(ctx->d_plane_internal, ctx->plane_internal.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_plane_cand_pos, ctx->plane_cand_pos.data(), np * 4, hipMemcpyHostToDevice, ctx->stream));
    ctx->plane_tabs_fresh = true;
  }
  if (N > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_fit_pw, point_w, N * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_fit_var, var, N * 72, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(devalloc::memcpy_async(ctx->d_fit_off, offsets, ((size_t)n_groups + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (plane_idx) HIPCHK(devalloc::memcpy_async(ctx->d_fit_idx, plane_idx, (size_t)n_groups * 4, hipMemcpyHostToDevice, ctx->stream));
  // small voxels (the UpdateVoxelMap case) go 8 to a wave, large ones (BuildVoxelMap) get a whole wave
  std::vector<int32_t> list((size_t)n_groups);
  int n_small = 0, n_big = 0;
  for (int g = 0; g < n_groups; g++) { if (offsets[g + 1] - offsets[g] <= 64) list[n_small++] = g; else list[n_groups - 1 - n_big++] = g; }
  HIPCHK(devalloc::memcpy_async(ctx->d_fit_list, list.data(), (size_t)n_groups * 4, hipMemcpyHostToDevice, ctx->stream));
  PlaneFitArgs a{};
  a.pw = ctx->d_fit_pw; a.var = ctx->d_fit_var; a.offsets = ctx->d_fit_off; a.planer_threshold = planer_threshold; a.out = ctx->d_fit_out;
  a.plane_idx = plane_idx ? ctx->d_fit_idx : nullptr; a.plane_internal = ctx->d_plane_internal; a.plane_cand_pos = ctx->d_plane_cand_pos;
  a.planes = ctx->d_planes; a.planes_hot = ctx->d_planes_hot; a.cand = ctx->d_cand; a.plane_aux = ctx->d_plane_aux; a.cand_aux = ctx->d_cand_aux;
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  const int tpb = FIT_WAVES * LIVO2_WAVE;
  if (n_small) { a.list = ctx->d_fit_list; a.n_list = n_small; hipLaunchKernelGGL(k_plane_fit<8>, dim3((n_small * 8 + tpb - 1) / tpb), dim3(tpb), 0, ctx->stream, a); }
  if (n_big) { a.list = ctx->d_fit_list + n_small; a.n_list = n_big; hipLaunchKernelGGL(k_plane_fit<64>, dim3((n_big * 64 + tpb - 1) / tpb), dim3(tpb), 0, ctx->stream, a); }
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  HIPCHK(hipGetLastError());
  HIPCHK(devalloc::memcpy_async(out, ctx->d_fit_out, (size_t)n_groups * sizeof(livo2_plane_fit), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->fit_kernel_us = 1e3 * ms;
  return LIVO2_OK;
}
double livo2_plane_fit_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->fit_kernel_us : 0.0; }

// ---- LiDAR -----------------------------------------------------------------------------------------------------------------
namespace {
// device buffers of one scan of up to n points
int scan_reserve(livo2_ctx *ctx, int n) {
  if (n > ctx->n_cap) {
    hipError_t e;
    if (ctx->d_x) { e = DFREE(ctx->d_xyz_aos); e = DFREE(ctx->d_x); e = DFREE(ctx->d_y); e = DFREE(ctx->d_z); e = DFREE(ctx->d_cb); e = DFREE(ctx->d_keys); e = DFREE(ctx->d_keys2); e = DFREE(ctx->d_idx); e = DFREE(ctx->d_perm); (void)e; }
    // headroom: live scans differ in size from frame to frame (the voxel-grid filter's output), and a re-allocation is nine hipFree / hipMalloc pairs that stall every
    // stream of the device — with an exact fit the C1-shaped frames of three contexts ran at 2 200 frames/s instead of 5 100 (profiles/r05_frame_api_probe.txt)
    int cap = std::max(n + n / 2, 4096);
    HIPCHK(DMALLOC((void **)&ctx->d_xyz_aos, (size_t)cap * 12)); HIPCHK(DMALLOC((void **)&ctx->d_x, (size_t)cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->d_y, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_z, (size_t)cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->d_cb, (size_t)cap * 48));
    HIPCHK(DMALLOC((void **)&ctx->d_keys, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_keys2, (size_t)cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->d_idx, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_perm, (size_t)cap * 4));
    ctx->n_cap = cap;
  }
  return LIVO2_OK;
}
int sort_reserve(livo2_ctx *ctx, size_t need) {
  if (need > ctx->sort_tmp_bytes) {
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (ctx->d_sort_tmp) HIPCHK(DFREE(ctx->d_sort_tmp));
    ctx->d_sort_tmp = nullptr;
    HIPCHK(DMALLOC(&ctx->d_sort_tmp, need + need / 2 + 256));
    ctx->sort_tmp_bytes = need + need / 2 + 256;
  }
  return LIVO2_OK;
}
// d_xyz_aos[0..n) holds feats_down_body: Morton order of the body-frame cells (cell = voxel_size), SoA gather, body covariance
int scan_pipeline(livo2_ctx *ctx, int n, const livo2_lidar_cfg *cfg) {
  ctx->n = n;
  ctx->lidar_block = lidar_block_for(n);
  const int grid = lidar_grid(std::max(n, 1), ctx->lidar_block);
  int rc = ensure(ctx, ctx->d_partials, ctx->partials_cap, std::max((size_t)grid * 32, (size_t)64));
  if (rc) return rc;
  // a new scan: the block lifetimes of the last one say nothing about it (identity order until this scan's first solve has run)
  rc = ensure(ctx, ctx->d_lpt_order, ctx->lpt_order_cap, 2 * (size_t)std::max(grid, 64)); if (rc) return rc;       // two orders: k_lidar_iteration fills one while its blocks read the other
  rc = ensure(ctx, ctx->d_lpt_cost, ctx->lpt_cost_cap, (size_t)std::max(grid, 64)); if (rc) return rc;
  HIPCHK(hipMemsetAsync(ctx->d_lpt_cost, 0, (size_t)grid * 4, ctx->stream));
  // arrival counters of k_lidar_iteration: zero here, and left at zero by the last arriver of every launch (a launch that was torn down half-way cannot poison the next scan)
  if (!ctx->d_lidar_tickets) HIPCHK(DMALLOC((void **)&ctx->d_lidar_tickets, LIDAR_TICKET_WORDS * 4));
  HIPCHK(hipMemsetAsync(ctx->d_lidar_tickets, 0, LIDAR_TICKET_WORDS * 4, ctx->stream));
  ctx->lpt_chunks = grid; ctx->lpt_valid = false;
  if (n > 0) {
    hipLaunchKernelGGL(k_morton_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_xyz_aos, n, (float)(1.0 / cfg->voxel_size), ctx->d_keys, ctx->d_idx);
    size_t need = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, need, ctx->d_keys, ctx->d_keys2, ctx->d_idx, ctx->d_perm, (size_t)n, 0, 30, ctx->stream));
    rc = sort_reserve(ctx, need); if (rc) return rc;
    size_t tmp_bytes = ctx->sort_tmp_bytes;
    HIPCHK(rocprim::radix_sort_pairs(ctx->d_sort_tmp, tmp_bytes, ctx->d_keys, ctx->d_keys2, ctx->d_idx, ctx->d_perm, (size_t)n, 0, 30, ctx->stream));
    hipLaunchKernelGGL(k_gather_xyz, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_xyz_aos, ctx->d_perm, n, ctx->d_x, ctx->d_y, ctx->d_z);
    const double deg2rad = cfg->deg2rad != 0.0 ? cfg->deg2rad : 0.017453293;
    // d_cb is [6][n] with row pitch n (not n_cap): the residual kernel indexes cb[e*n + i]
    hipLaunchKernelGGL(k_body_cov, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_x, ctx->d_y, ctx->d_z, n, (float)cfg->dept_err, (float)cfg->beam_err,
                       deg2rad, ctx->d_cb);
    HIPCHK(hipGetLastError());
  }
  return LIVO2_OK;
}
} // namespace

// ---- device-resident VoxelMap ------------------------------------------------------------------------------------------------------------
int livo2_map_tree_create(livo2_ctx *ctx, const livo2_map_tree_cfg *cfg) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!cfg || !(cfg->voxel_size > 0) || cfg->max_layer < 0 || cfg->max_layer > LIVO2_MAX_LAYER || cfg->max_points_num < 1 || cfg->max_points_num > LIVO2_MAX_POINTS_NUM || cfg->max_roots < 1)
    return fail(ctx, LIVO2_ERR_INVALID, "bad map tree cfg (max_layer in [0,LIVO2_MAX_LAYER], max_points_num in [1,LIVO2_MAX_POINTS_NUM], max_roots >= 1)");
  const int slab = std::max(MT_SLAB_MIN, cfg->max_points_num + 2);
  for (int k = 0; k < 5; k++) if (cfg->layer_init_num[k] < 1 || cfg->layer_init_num[k] > slab - 2) return fail(ctx, LIVO2_ERR_INVALID, "layer_init_num out of [1, max(50, max_points_num)]");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  hipError_t e;
  e = hipSuccess;
  free_map_arrays(ctx);
  if (ctx->mt.nodes) { e = DFREE(ctx->mt.nodes); e = DFREE(ctx->mt.pool_pw); e = DFREE(ctx->mt.pool_var); e = DFREE(ctx->mt.counters); e = DFREE(ctx->mt.dirty_list); e = DFREE(ctx->mt.overflow_list); }
  (void)e;
  ctx->has_map = false; ctx->tree_mode = false; ctx->mt = MapTreeArgs{};
  MapTreeArgs &m = ctx->mt;
  const long long R = cfg->max_roots;
  m.cap_nodes = (int32_t)std::min<long long>(cfg->max_nodes > 0 ? cfg->max_nodes : 3 * R, INT32_MAX / 2);
  m.cap_planes = (int32_t)std::min<long long>(cfg->max_planes > 0 ? cfg->max_planes : 2 * R, (1 << CAND_LAYER_SHIFT) - 1);
  m.cap_points = (int32_t)std::min<long long>(cfg->max_points > 0 ? cfg->max_points : (long long)(80 * std::max(MT_SLAB_MIN, cfg->max_points_num + 2) / MT_SLAB_MIN) * R, INT32_MAX / 16);
  m.cap_cand = (int32_t)std::min<long long>(cfg->max_cand > 0 ? cfg->max_cand : R, INT32_MAX / 64);
  m.cap_overflow = (int32_t)std::max<long long>(1024, R / 4);
  uint32_t cap = 64;
  while ((long long)cap < 8 * R) cap <<= 1;
  HIPCHK(DMALLOC((void **)&ctx->d_slots, (size_t)cap * sizeof(RootSlot)));
  HIPCHK(DMALLOC((void **)&ctx->d_planes, (size_t)m.cap_planes * PLANE_REC_DOUBLES * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_cand, (size_t)m.cap_cand * PLANE_HOT_DOUBLES * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_cand_aux, (size_t)m.cap_cand * sizeof(PlaneAux)));
  HIPCHK(DMALLOC((void **)&ctx->d_planes_hot, (size_t)m.cap_planes * PLANE_HOT_DOUBLES * 8));
  HIPCHK(DMALLOC((void **)&ctx->d_plane_aux, (size_t)m.cap_planes * sizeof(PlaneAux)));
  HIPCHK(DMALLOC((void **)&m.nodes, (size_t)m.cap_nodes * sizeof(DevNode)));
  HIPCHK(DMALLOC((void **)&m.pool_pw, (size_t)m.cap_points * 24));
  HIPCHK(DMALLOC((void **)&m.pool_var, (size_t)m.cap_points * 72));
  // counters + the free stacks behind them (map_tree_kernels.hpp: mt_free_nodes ... mt_pending_slabs)
  HIPCHK(DMALLOC((void **)&m.counters, ((size_t)MTC_TOTAL + m.cap_nodes + m.cap_planes + ((size_t)m.cap_points / slab + 1)) * 4));
  m.slab = slab;
  HIPCHK(DMALLOC((void **)&m.dirty_list, (size_t)m.cap_nodes * 4));
  HIPCHK(DMALLOC((void **)&m.overflow_list, (size_t)m.cap_overflow * 4));
  {
    std::vector<RootSlot> empty(cap);
    for (auto &sl : empty) { sl.val = -1; sl.kx = sl.ky = sl.kz = MT_NO_KEY; sl.pad = -1; }
    HIPCHK(hipMemcpy(ctx->d_slots, empty.data(), (size_t)cap * sizeof(RootSlot), hipMemcpyHostToDevice));
  }
  HIPCHK(hipMemset(m.counters, 0, MTC_TOTAL * 4));
  ctx->mt_last_slide[0] = ctx->mt_last_slide[1] = ctx->mt_last_slide[2] = 0.0;          // VoxelMapManager::last_slide_position (voxel_map.h:209)
  HIPCHK(hipMemset(ctx->d_planes, 0, (size_t)m.cap_planes * PLANE_REC_DOUBLES * 8));
  HIPCHK(hipMemset(ctx->d_cand, 0, (size_t)m.cap_cand * PLANE_HOT_DOUBLES * 8));
  HIPCHK(hipMemset(ctx->d_cand_aux, 0, (size_t)m.cap_cand * sizeof(PlaneAux)));
  HIPCHK(hipMemset(ctx->d_planes_hot, 0, (size_t)m.cap_planes * PLANE_HOT_DOUBLES * 8));
  HIPCHK(hipMemset(ctx->d_plane_aux, 0, (size_t)m.cap_planes * sizeof(PlaneAux)));
  m.planes = ctx->d_planes; m.planes_hot = ctx->d_planes_hot; m.cand = ctx->d_cand; m.plane_aux = ctx->d_plane_aux; m.cand_aux = ctx->d_cand_aux; m.slots = ctx->d_slots;
  m.mask = cap - 1; m.seed1 = 0x243f6a88u; m.seed2 = 0x85a308d3u;
  m.voxel_size_d = cfg->voxel_size; m.voxel_size_f = (float)cfg->voxel_size; m.planer_threshold = (float)cfg->planer_threshold;
  m.max_layer = cfg->max_layer; m.max_points_num = cfg->max_points_num; m.update_size_threshold = 5;      // VoxelOctoTree ctor (voxel_map.h:159)
  for (int k = 0; k <= LIVO2_MAX_LAYER; k++) m.layer_init_num[k] = cfg->layer_init_num[k < 5 ? k : 4];
  set_map_view(ctx);
  ctx->map.mask = m.mask; ctx->map.seed1 = m.seed1; ctx->map.seed2 = m.seed2;
  ctx->map.n_planes = m.cap_planes;
  ctx->mt_cfg = *cfg;
  ctx->plane_internal.clear(); ctx->plane_orig.clear(); ctx->plane_cand_pos.clear();
  if (!ctx->mt_nseg) HIPCHK(DMALLOC((void **)&ctx->mt_nseg, 64));
  if (!ctx->mt_state) HIPCHK(DMALLOC((void **)&ctx->mt_state, sizeof(livo2_state)));
  ctx->has_map = true; ctx->tree_mode = true;
  return LIVO2_OK;
}

namespace {
// sort by root voxel, segment, roots, octree update, emit — on points already in mt_in_pw / mt_in_var
int map_tree_run(livo2_ctx *ctx, int n, int build) {
  MapTreeArgs &m = ctx->mt;
  int rc;
  if ((rc = ensure(ctx, ctx->mt_keys, ctx->mt_keys_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_keys2, ctx->mt_keys2_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_idx, ctx->mt_idx_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_order, ctx->mt_order_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_head, ctx->mt_head_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_slot, ctx->mt_slot_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->mt_seg_begin, ctx->mt_seg_begin_cap, (size_t)n + 2))) return rc;
  if ((rc = ensure(ctx, ctx->mt_seg_root, ctx->mt_seg_root_cap, (size_t)n + 1))) return rc;
  HIPCHK(hipMemsetAsync(m.counters + MTC_OVERFLOW, 0, 3 * 4, ctx->stream));          // overflow, error, dirty
  HIPCHK(hipMemsetAsync(ctx->mt_nseg, 0, 4, ctx->stream));
  if (n == 0) return LIVO2_OK;
  const int nb = (n + 255) / 256;
  hipLaunchKernelGGL(k_mt_keys, dim3(nb), dim3(256), 0, ctx->stream, ctx->mt_in_pw, n, m.voxel_size_f, ctx->mt_keys, ctx->mt_idx, m.counters);
  size_t need = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, need, ctx->mt_keys, ctx->mt_keys2, ctx->mt_idx, ctx->mt_order, (size_t)n, 0, 63, ctx->stream));
  rc = sort_reserve(ctx, need); if (rc) return rc;
  size_t tmp_bytes = ctx->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(ctx->d_sort_tmp, tmp_bytes, ctx->mt_keys, ctx->mt_keys2, ctx->mt_idx, ctx->mt_order, (size_t)n, 0, 63, ctx->stream));
  hipLaunchKernelGGL(k_mt_heads, dim3(nb), dim3(256), 0, ctx->stream, ctx->mt_keys2, n, ctx->mt_head);
  {
    size_t scan_need = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, scan_need, ctx->mt_head, ctx->mt_slot, 0, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
    rc = sort_reserve(ctx, scan_need); if (rc) return rc;
    size_t scan_bytes = ctx->sort_tmp_bytes;
    HIPCHK(rocprim::exclusive_scan(ctx->d_sort_tmp, scan_bytes, ctx->mt_head, ctx->mt_slot, 0, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
  }
  hipLaunchKernelGGL(k_mt_segments, dim3(nb), dim3(256), 0, ctx->stream, ctx->mt_head, ctx->mt_slot, n, ctx->mt_seg_begin, ctx->mt_nseg);
  MapTreeArgs a = m;
  a.in_pw = ctx->mt_in_pw; a.in_var = ctx->mt_in_var; a.order = ctx->mt_order; a.skeys = ctx->mt_keys2; a.seg_head = ctx->mt_head; a.seg_slot = ctx->mt_slot;
  a.seg_begin = ctx->mt_seg_begin; a.seg_root = ctx->mt_seg_root; a.n = n; a.build = build ? 1 : 0;
  // the segment count stays on the device: grids are sized for the worst case (one segment per point), surplus threads leave at once
  // recycling variants only when there is something to recycle and a pool is more than half used (the plain kernels are ~40 % faster: map_tree_kernels.hpp)
  const bool recycle = m.may_pop != 0 && ctx->mt_pool_pressure;
  if (recycle) hipLaunchKernelGGL(k_mt_roots<true>, dim3(nb), dim3(256), 0, ctx->stream, a, ctx->mt_nseg);
  else hipLaunchKernelGGL(k_mt_roots<false>, dim3(nb), dim3(256), 0, ctx->stream, a, ctx->mt_nseg);
  hipLaunchKernelGGL(k_mt_overflow, dim3(1), dim3(64), 0, ctx->stream, a);
  if (recycle) hipLaunchKernelGGL(k_mt_update<true>, dim3((n * MT_LPG + 255) / 256), dim3(256), 0, ctx->stream, a, ctx->mt_nseg);
  else hipLaunchKernelGGL(k_mt_update<false>, dim3((n * MT_LPG + 255) / 256), dim3(256), 0, ctx->stream, a, ctx->mt_nseg);
  hipLaunchKernelGGL(k_mt_collect, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, a);        // regions of the nodes that froze in this update -> free stack (at most n roots are dirty)
  hipLaunchKernelGGL(k_mt_emit, dim3((n * MT_LPG + 255) / 256), dim3(256), 0, ctx->stream, a);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}
// ---- the pools of the device tree grow on demand --------------------------------------------------------------------------------------------
// After every update the host looks at the pool counters (it reads them anyway).  A pool that is more than `MT_GROW_AT` full is doubled before the next update:
// new allocation, device-to-device copy, pointers swapped (the ctx stream is idle here).  An update can therefore only fail with a capacity error if ONE frame needs
// more than the free part of a pool — i.e. more than (1 - MT_GROW_AT) of everything the map has accumulated so far.  The root hash table does not grow (its
// size is fixed by max_roots; 8 buckets per root).  Candidate ranges are first re-packed (all lists re-emitted back to back) and only grown if that is not enough.
#define MT_GROW_AT 0.6
int map_tree_relayout_counters(livo2_ctx *ctx, int new_nodes, int new_planes, int new_points) {
  MapTreeArgs &m = ctx->mt;
  const size_t new_total = (size_t)MTC_TOTAL + new_nodes + new_planes + ((size_t)new_points / m.slab + 1);
  int32_t *q = nullptr;
  HIPCHK(DMALLOC((void **)&q, new_total * 4));
  HIPCHK(devalloc::memcpy_async(q, m.counters, (size_t)MTC_TOTAL * 4, hipMemcpyDeviceToDevice, ctx->stream));
  // the three free stacks move to their new offsets (copied whole: their tops are the counters MTC_FREE_*)
  HIPCHK(devalloc::memcpy_async(q + MTC_TOTAL, m.counters + MTC_TOTAL, (size_t)m.cap_nodes * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(q + MTC_TOTAL + new_nodes, m.counters + MTC_TOTAL + m.cap_nodes, (size_t)m.cap_planes * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(q + MTC_TOTAL + new_nodes + new_planes, m.counters + MTC_TOTAL + m.cap_nodes + m.cap_planes, ((size_t)m.cap_points / m.slab + 1) * 4, hipMemcpyDeviceToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(DFREE(m.counters));
  m.counters = q;
  return LIVO2_OK;
}
int map_tree_grow(livo2_ctx *ctx, const int32_t *c) {
  MapTreeArgs &m = ctx->mt;
  int rc;
  const auto over = [](long long used, long long cap) { return (double)used > MT_GROW_AT * (double)cap; };
  // candidate ranges: re-pack first (every root re-emits its list into an empty pool), and only if the PACKED lists still fill more than MT_GROW_AT of the pool, or do
  // not fit at all, grow it.  A pool that cannot hold the lists after four doublings leaves MTE_CAND set: the caller fails the frame instead of running residuals
  // against slots that point into a pool nobody filled (advisor, round 3).
  if (over(c[MTC_CAND], m.cap_cand)) {
    int32_t err = 0, used = 0;
    for (int attempt = 0; attempt < 5; attempt++) {
      if (attempt > 0) {                                                  // the packed lists did not fit: double the pool (nothing to preserve: every list is re-emitted)
        const int nc = (int)std::min<long long>(2LL * m.cap_cand, INT32_MAX / 64);
        if (nc <= m.cap_cand) break;
        if ((rc = grow_array(ctx, ctx->d_cand, 0, (size_t)nc * PLANE_HOT_DOUBLES, false))) return rc;
        if ((rc = grow_array(ctx, ctx->d_cand_aux, 0, (size_t)nc, false))) return rc;
        m.cap_cand = nc; m.cand = ctx->d_cand; m.cand_aux = ctx->d_cand_aux;
        err &= ~MTE_CAND;
        HIPCHK(devalloc::memcpy_async(m.counters + MTC_ERROR, &err, 4, hipMemcpyHostToDevice, ctx->stream));
      }
      HIPCHK(hipMemsetAsync(m.counters + MTC_CAND, 0, 4, ctx->stream));
      HIPCHK(hipMemsetAsync(m.counters + MTC_DIRTY, 0, 4, ctx->stream));
      MapTreeArgs a = m;
      hipLaunchKernelGGL(k_mt_all_roots_dirty, dim3((m.mask + 256) / 256), dim3(256), 0, ctx->stream, a);
      int32_t nd = 0;
      HIPCHK(devalloc::memcpy_async(&nd, m.counters + MTC_DIRTY, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (nd > 0) hipLaunchKernelGGL(k_mt_emit, dim3(((size_t)nd * MT_LPG + 255) / 256), dim3(256), 0, ctx->stream, a);
      HIPCHK(devalloc::memcpy_async(&err, m.counters + MTC_ERROR, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(devalloc::memcpy_async(&used, m.counters + MTC_CAND, 4, hipMemcpyDeviceToHost, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (!(err & MTE_CAND)) break;
    }
    HIPCHK(hipMemsetAsync(m.counters + MTC_DIRTY, 0, 4, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    set_map_view(ctx);
    ctx->mt_grow_events++;
    if (err & MTE_CAND) return fail(ctx, LIVO2_ERR_RANGE, "device map tree: the candidate lists do not fit the candidate pool even after re-packing and four doublings");
    if (over(used, m.cap_cand)) {                                         // packed and still more than MT_GROW_AT full: double, keeping the lists
      const int nc = (int)std::min<long long>(2LL * m.cap_cand, INT32_MAX / 64);
      if (nc > m.cap_cand) {
        if ((rc = grow_array(ctx, ctx->d_cand, (size_t)used * PLANE_HOT_DOUBLES, (size_t)nc * PLANE_HOT_DOUBLES, false))) return rc;
        if ((rc = grow_array(ctx, ctx->d_cand_aux, (size_t)used, (size_t)nc, false))) return rc;
        m.cap_cand = nc; m.cand = ctx->d_cand; m.cand_aux = ctx->d_cand_aux;
        set_map_view(ctx);
      }
    }
  }
  const bool gn = over(c[MTC_NODES], m.cap_nodes), gp = over(c[MTC_POINTS], m.cap_points), gl = over(c[MTC_PLANES], m.cap_planes);
  if (gn || gp || gl) {
    const int nn = gn ? (int)std::min<long long>(2LL * m.cap_nodes, INT32_MAX / 2) : m.cap_nodes;
    const int np = gp ? (int)std::min<long long>(2LL * m.cap_points, INT32_MAX / 16) : m.cap_points;
    const int nl = gl ? (int)std::min<long long>(2LL * m.cap_planes, (1 << CAND_LAYER_SHIFT) - 1) : m.cap_planes;
    if ((rc = map_tree_relayout_counters(ctx, nn, nl, np))) return rc;
    if (gn) {
      if ((rc = grow_array(ctx, m.nodes, (size_t)m.cap_nodes, (size_t)nn, false))) return rc;
      if ((rc = grow_array(ctx, m.dirty_list, (size_t)m.cap_nodes, (size_t)nn, false))) return rc;
    }
    if (gp) {
      if ((rc = grow_array(ctx, m.pool_pw, (size_t)m.cap_points * 3, (size_t)np * 3, false))) return rc;
      if ((rc = grow_array(ctx, m.pool_var, (size_t)m.cap_points * 9, (size_t)np * 9, false))) return rc;
    }
    if (gl) {
      if ((rc = grow_array(ctx, ctx->d_planes, (size_t)m.cap_planes * PLANE_REC_DOUBLES, (size_t)nl * PLANE_REC_DOUBLES, true))) return rc;
      if ((rc = grow_array(ctx, ctx->d_planes_hot, (size_t)m.cap_planes * PLANE_HOT_DOUBLES, (size_t)nl * PLANE_HOT_DOUBLES, true))) return rc;
      if ((rc = grow_array(ctx, ctx->d_plane_aux, (size_t)m.cap_planes, (size_t)nl, true))) return rc;
    }
    m.cap_nodes = nn; m.cap_points = np; m.cap_planes = nl;
    m.planes = ctx->d_planes; m.planes_hot = ctx->d_planes_hot; m.plane_aux = ctx->d_plane_aux;
    set_map_view(ctx);
    ctx->map.n_planes = m.cap_planes;
    ctx->mt_grow_events++;
  }
  return LIVO2_OK;
}

int map_tree_finish(livo2_ctx *ctx) {
  int32_t c[MTC_COUNT], fr[5];
  HIPCHK(devalloc::memcpy_async(c, ctx->mt.counters, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(fr, ctx->mt.counters + MTC_FREE_NODES, sizeof(fr), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  // what the next update may pop: the stacks as they are + the regions this update's frozen nodes released (joined to the stack before that update starts)
  ctx->mt.may_pop = (fr[0] > 0 ? 1 : 0) | (fr[1] > 0 ? 2 : 0) | (fr[2] > 0 ? 4 : 0);
  {
    static const bool always = [] { const char *e = std::getenv("LIVO2_MAP_RECYCLE"); return e && std::atoi(e) != 0; }();
    ctx->mt_pool_pressure = always || 2 * (long long)c[MTC_NODES] > ctx->mt.cap_nodes || 2 * (long long)c[MTC_POINTS] > ctx->mt.cap_points || 2 * (long long)c[MTC_PLANES] > ctx->mt.cap_planes;
  }
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->mt_kernel_us = 1e3 * ms;
  if (!c[MTC_ERROR]) { const int rc = map_tree_grow(ctx, c); if (rc) return rc; }
  if (c[MTC_ERROR]) {
    // A frame that exhausted a pool: the allocators rolled their counters back, so the pool is still (nearly) full and every later frame would fail the same way
    // (advisor, round 3).  Double every pool whose bit is set NOW, clear those bits, and report this frame as dropped: the tree stays usable for the next one.
    const int32_t cap_bits = c[MTC_ERROR] & (MTE_NODES | MTE_POINTS | MTE_PLANES | MTE_CAND);
    int grown = 0;
    if (cap_bits) {
      int32_t forced[MTC_COUNT];
      std::memcpy(forced, c, sizeof(forced));
      if (cap_bits & MTE_NODES) forced[MTC_NODES] = ctx->mt.cap_nodes;
      if (cap_bits & MTE_POINTS) forced[MTC_POINTS] = ctx->mt.cap_points;
      if (cap_bits & MTE_PLANES) forced[MTC_PLANES] = ctx->mt.cap_planes;
      if (cap_bits & MTE_CAND) forced[MTC_CAND] = ctx->mt.cap_cand;
      const int32_t rest = c[MTC_ERROR] & ~cap_bits;
      HIPCHK(devalloc::memcpy_async(ctx->mt.counters + MTC_ERROR, &rest, 4, hipMemcpyHostToDevice, ctx->stream));
      HIPCHK(hipStreamSynchronize(ctx->stream));
      const int rc = map_tree_grow(ctx, forced);
      grown = rc == LIVO2_OK;
    }
    char msg[320];                                            // (fail() copies it into the ctx's own string)
    std::snprintf(msg, sizeof(msg), "device map tree: capacity / range error bits 0x%x (1 nodes, 2 points, 4 planes, 8 candidate lists, 16 hash table, 32 voxel key range, 64 node region)%s",
                  c[MTC_ERROR], grown ? "; this frame's update is incomplete, the exhausted pools were doubled for the next one" : "");
    return fail(ctx, LIVO2_ERR_RANGE, msg);
  }
  return LIVO2_OK;
}
} // namespace

int livo2_map_tree_update(livo2_ctx *ctx, const double *point_w, const double *var, int32_t n, int32_t build) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_map_tree_create has not been called");
  if (n < 0 || (n > 0 && (!point_w || !var))) return fail(ctx, LIVO2_ERR_INVALID, "bad input_points");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc;
  if ((rc = ensure(ctx, ctx->mt_in_pw, ctx->mt_in_pw_cap, (size_t)std::max(n, 1) * 3))) return rc;
  if ((rc = ensure(ctx, ctx->mt_in_var, ctx->mt_in_var_cap, (size_t)std::max(n, 1) * 9))) return rc;
  if (n > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->mt_in_pw, point_w, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->mt_in_var, var, (size_t)n * 72, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  ctx->mt_pv_n = n;
  rc = map_tree_run(ctx, n, build); if (rc) return rc;
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  return map_tree_finish(ctx);
}

int livo2_map_tree_update_from_scan(livo2_ctx *ctx, const livo2_state *state, const livo2_lidar_cfg *cfg, int32_t build) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_map_tree_create has not been called");
  if (!state) return fail(ctx, LIVO2_ERR_INVALID, "state is NULL");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  if (!ctx->has_scan) return fail(ctx, LIVO2_ERR_NO_SCAN, "livo2_lidar_set_scan has not been called");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int n = ctx->n;
  if ((rc = ensure(ctx, ctx->mt_in_pw, ctx->mt_in_pw_cap, (size_t)std::max(n, 1) * 3))) return rc;
  if ((rc = ensure(ctx, ctx->mt_in_var, ctx->mt_in_var_cap, (size_t)std::max(n, 1) * 9))) return rc;
  HIPCHK(devalloc::memcpy_async(ctx->mt_state, state, sizeof(livo2_state), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  if (n > 0) {
    MapPvArgs p{};
    p.x = ctx->d_x; p.y = ctx->d_y; p.z = ctx->d_z; p.cb = ctx->d_cb; p.perm = ctx->d_perm; p.n = n;
    std::memcpy(p.ER, cfg->extR, 72); std::memcpy(p.Et, cfg->extT, 24);
    p.out_pw = ctx->mt_in_pw; p.out_var = ctx->mt_in_var;
    hipLaunchKernelGGL(k_mt_pv_from_scan, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, p, ctx->mt_state);
  }
  ctx->mt_pv_n = n;
  rc = map_tree_run(ctx, n, build); if (rc) return rc;
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  return map_tree_finish(ctx);
}

int livo2_map_tree_read_pv(livo2_ctx *ctx, double *point_w, double *var, int32_t capacity, int32_t *n) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (ctx->mt_pv_n < 0) return fail(ctx, LIVO2_ERR_NO_SCAN, "no livo2_map_tree_update[_from_scan] since the last livo2_lidar_set_scan");
  if (n) *n = ctx->mt_pv_n;
  if (capacity < ctx->mt_pv_n) return fail(ctx, LIVO2_ERR_INVALID, "capacity smaller than the pv_list");
  HIPCHK(hipSetDevice(ctx->device));
  if (ctx->mt_pv_n > 0) {
    if (point_w) HIPCHK(devalloc::memcpy_async(point_w, ctx->mt_in_pw, (size_t)ctx->mt_pv_n * 24, hipMemcpyDeviceToHost, ctx->stream));
    if (var) HIPCHK(devalloc::memcpy_async(var, ctx->mt_in_var, (size_t)ctx->mt_pv_n * 72, hipMemcpyDeviceToHost, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LIVO2_OK;
}

int livo2_map_tree_stats(livo2_ctx *ctx, int32_t *counts) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode || !counts) return fail(ctx, LIVO2_ERR_INVALID, "no device map tree / counts is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(devalloc::memcpy_async(counts, ctx->mt.counters, MTC_COUNT * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LIVO2_OK;
}
double livo2_map_tree_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->mt_kernel_us : 0.0; }

// VoxelMapManager::mapSliding (voxel_map.cpp:924-948) + clearMemOutOfMap (950-972) on the device tree
int livo2_map_tree_slide(livo2_ctx *ctx, const double *position_last, double sliding_thresh, int32_t half_map_size, int32_t *removed, int32_t *free_counts) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "no device map tree");
  if (!position_last) return fail(ctx, LIVO2_ERR_INVALID, "position_last is NULL");
  if (half_map_size < 0 || !(sliding_thresh >= 0) || !std::isfinite(position_last[0]) || !std::isfinite(position_last[1]) || !std::isfinite(position_last[2]))
    return fail(ctx, LIVO2_ERR_INVALID, "bad sliding arguments (half_map_size >= 0, sliding_thresh >= 0, finite position)");
  HIPCHK(hipSetDevice(ctx->device));
  const double dx = position_last[0] - ctx->mt_last_slide[0], dy = position_last[1] - ctx->mt_last_slide[1], dz = position_last[2] - ctx->mt_last_slide[2];
  int32_t c[5];                                                  // free nodes / planes / slabs, removed, pending
  bool slid = false;
  if (std::sqrt((dx * dx + dy * dy) + dz * dz) < sliding_thresh) {          // (position_last_ - last_slide_position).norm() < sliding_thresh: nothing happens
  } else {
    slid = true;
    for (int k = 0; k < 3; k++) ctx->mt_last_slide[k] = position_last[k];
    int64_t loc[3];
    for (int j = 0; j < 3; j++) {                               // float loc = position_last_[j] / max_voxel_size_; negatives one lower; truncation
      float l = (float)(position_last[j] / ctx->mt_cfg.voxel_size);
      if (l < 0) l = (float)((double)l - 1.0);
      loc[j] = (int64_t)l;
    }
    // clearMemOutOfMap takes `const int &`: the int64 sums are narrowed at the call
    SlideBox b = {(int)(loc[0] + half_map_size), (int)(loc[0] - half_map_size), (int)(loc[1] + half_map_size), (int)(loc[1] - half_map_size),
                  (int)(loc[2] + half_map_size), (int)(loc[2] - half_map_size)};
    HIPCHK(hipMemsetAsync(ctx->mt.counters + MTC_REMOVED, 0, 4, ctx->stream));
    hipLaunchKernelGGL(k_mt_slide, dim3((ctx->mt.mask + 256) / 256), dim3(256), 0, ctx->stream, ctx->mt, b);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(devalloc::memcpy_async(c, ctx->mt.counters + MTC_FREE_NODES, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->mt.may_pop = (c[0] > 0 ? 1 : 0) | (c[1] > 0 ? 2 : 0) | (c[2] > 0 ? 4 : 0);
  if (removed) *removed = slid ? c[3] : -1;
  if (free_counts) { free_counts[0] = c[0]; free_counts[1] = c[1]; free_counts[2] = c[2]; }
  return LIVO2_OK;
}

int livo2_map_tree_read_planes(livo2_ctx *ctx, const int32_t *rows, int32_t n, double *normal, double *center, double *plane_var, float *d, float *radius, int32_t *layer) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "no device map tree");
  if (n < 0 || (n > 0 && !rows)) return fail(ctx, LIVO2_ERR_INVALID, "bad rows");
  if (n == 0) return LIVO2_OK;
  for (int i = 0; i < n; i++) if (rows[i] < 0 || rows[i] >= ctx->mt.cap_planes) return fail(ctx, LIVO2_ERR_INVALID, "plane row out of range");
  HIPCHK(hipSetDevice(ctx->device));
  // ctx-owned, growable staging (a hipMalloc / hipFree pair per call synchronises the whole device: every other chain on the GPU would stall)
  int rc = ensure(ctx, ctx->mt_rp_rows, ctx->mt_rp_rows_cap, (size_t)n); if (rc) return rc;
  rc = ensure(ctx, ctx->mt_rp_out, ctx->mt_rp_out_cap, (size_t)n * PLANE_REC_DOUBLES); if (rc) return rc;
  int32_t *d_rows = ctx->mt_rp_rows; double *d_out = ctx->mt_rp_out;
  HIPCHK(devalloc::memcpy_async(d_rows, rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_mt_gather_planes, dim3((n * 32 + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_planes, d_rows, n, d_out);
  std::vector<double> recs((size_t)n * PLANE_REC_DOUBLES);
  HIPCHK(devalloc::memcpy_async(recs.data(), d_out, recs.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  for (int p = 0; p < n; p++) {
    const double *rec = &recs[(size_t)p * PLANE_REC_DOUBLES];
    if (normal) for (int k = 0; k < 3; k++) normal[(size_t)p * 3 + k] = rec[k];
    if (center) for (int k = 0; k < 3; k++) center[(size_t)p * 3 + k] = rec[3 + k];
    if (plane_var) { int q = 6; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { plane_var[(size_t)p * 36 + a * 6 + b] = rec[q]; plane_var[(size_t)p * 36 + b * 6 + a] = rec[q]; q++; } }
    float dr[2]; std::memcpy(dr, &rec[27], 8);
    int32_t meta[2]; std::memcpy(meta, &rec[28], 8);
    if (d) d[p] = dr[0];
    if (radius) radius[p] = dr[1];
    if (layer) layer[p] = meta[0];
  }
  return LIVO2_OK;
}

int livo2_map_tree_export(livo2_ctx *ctx, int64_t *root_key, int32_t *root_node, double *root_center, float *root_quarter, int32_t *node_plane, int32_t *node_child,
                          double *plane_normal, double *plane_center, double *plane_var, float *plane_d, float *plane_radius, int32_t *node_temp) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "no device map tree");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));                  // an update enqueued on the ctx stream must have finished before the blocking copies below read the tree
  int32_t c[MTC_COUNT];
  HIPCHK(hipMemcpy(c, ctx->mt.counters, sizeof(c), hipMemcpyDeviceToHost));
  const int nn = std::min(c[MTC_NODES], ctx->mt.cap_nodes), np = std::min(c[MTC_PLANES], ctx->mt.cap_planes);
  std::vector<DevNode> nodes((size_t)std::max(nn, 1));
  std::vector<double> recs((size_t)std::max(np, 1) * PLANE_REC_DOUBLES);
  if (nn) HIPCHK(hipMemcpy(nodes.data(), ctx->mt.nodes, (size_t)nn * sizeof(DevNode), hipMemcpyDeviceToHost));
  if (np) HIPCHK(hipMemcpy(recs.data(), ctx->d_planes, (size_t)np * PLANE_REC_DOUBLES * 8, hipMemcpyDeviceToHost));
  int r = 0;
  for (int i = 0; i < nn; i++) {
    const DevNode &nd = nodes[i];
    if (node_plane) node_plane[i] = nd.is_plane ? nd.plane : -1;
    if (node_child) for (int k = 0; k < 8; k++) node_child[(size_t)i * 8 + k] = nd.child[k];
    if (node_temp) node_temp[i] = nd.n_temp;
    if (nd.layer == 0 && nd.root == i) {
      if (root_key) for (int k = 0; k < 3; k++) root_key[(size_t)r * 3 + k] = nd.key[k];
      if (root_node) root_node[r] = i;
      if (root_center) for (int k = 0; k < 3; k++) root_center[(size_t)r * 3 + k] = nd.center[k];
      if (root_quarter) root_quarter[r] = nd.quarter;
      r++;
    }
  }
  for (int p = 0; p < np; p++) {
    const double *rec = &recs[(size_t)p * PLANE_REC_DOUBLES];
    if (plane_normal) for (int k = 0; k < 3; k++) plane_normal[(size_t)p * 3 + k] = rec[k];
    if (plane_center) for (int k = 0; k < 3; k++) plane_center[(size_t)p * 3 + k] = rec[3 + k];
    if (plane_var) { int q = 6; for (int a = 0; a < 6; a++) for (int b = a; b < 6; b++) { plane_var[(size_t)p * 36 + a * 6 + b] = rec[q]; plane_var[(size_t)p * 36 + b * 6 + a] = rec[q]; q++; } }
    float dr[2]; std::memcpy(dr, &rec[27], 8);
    if (plane_d) plane_d[p] = dr[0];
    if (plane_radius) plane_radius[p] = dr[1];
  }
  return LIVO2_OK;
}


int livo2_lidar_set_scan(livo2_ctx *ctx, const float *xyz, int32_t n, const livo2_lidar_cfg *cfg) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n < 0 || (n > 0 && !xyz)) return fail(ctx, LIVO2_ERR_INVALID, "bad scan");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  if (n > ctx->n_cap) HIPCHK(hipStreamSynchronize(ctx->stream));      // the scan buffers are about to be re-allocated
  ctx->mt_pv_n = -1;                                                   // the resident pv_list belonged to the previous scan
  rc = scan_reserve(ctx, n); if (rc) return rc;
  if (n > 0) {
    // the caller's (pageable) array goes through one of two pinned staging blocks: no stream synchronisation at entry or exit, the H2D of this scan
    // overlaps whatever the stream is still doing; a block is rewritten only after the copy that read it last has completed
    const int k = ctx->scan_stage_next; ctx->scan_stage_next ^= 1;
    const size_t bytes = (size_t)n * 12;
    if (ctx->scan_stage_used[k]) HIPCHK(hipEventSynchronize(ctx->scan_stage_ev[k]));
    if (bytes > ctx->scan_stage_cap[k]) {
      if (ctx->scan_stage[k]) HIPCHK(hipHostFree(ctx->scan_stage[k]));
      ctx->scan_stage[k] = nullptr; ctx->scan_stage_cap[k] = 0;
      HIPCHK(hipHostMalloc(&ctx->scan_stage[k], 2 * bytes));                        // (hipHostMalloc / hipHostFree synchronise the device: grow rarely)
      ctx->scan_stage_cap[k] = 2 * bytes;
    }
    if (!ctx->scan_stage_ev[k]) HIPCHK(hipEventCreateWithFlags(&ctx->scan_stage_ev[k], hipEventDisableTiming));
    std::memcpy(ctx->scan_stage[k], xyz, bytes);
    HIPCHK(devalloc::memcpy_async(ctx->d_xyz_aos, ctx->scan_stage[k], bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->scan_stage_ev[k], ctx->stream));
    ctx->scan_stage_used[k] = true;
  }
  rc = scan_pipeline(ctx, n, cfg); if (rc) return rc;
  ctx->has_scan = true; ctx->mt_pv_n = -1;
  return LIVO2_OK;
}

namespace {
// buffers of the pre-stage for n raw points and n_poses IMU poses
int preprocess_reserve(livo2_ctx *ctx, int n, int n_poses) {
  int rc = scan_reserve(ctx, n); if (rc) return rc;
  rc = ensure(ctx, ctx->d_raw, ctx->raw_cap, std::max((size_t)n * 3, (size_t)3)); if (rc) return rc;
  rc = ensure(ctx, ctx->d_curv, ctx->curv_cap, std::max((size_t)n, (size_t)1)); if (rc) return rc;
  rc = ensure(ctx, ctx->d_poses, ctx->poses_cap, std::max((size_t)n_poses * 22, (size_t)22)); if (rc) return rc;
  rc = ensure(ctx, ctx->d_vg_head, ctx->vg_head_cap, std::max((size_t)n, (size_t)1)); if (rc) return rc;
  rc = ensure(ctx, ctx->d_vg_slot, ctx->vg_slot_cap, std::max((size_t)n, (size_t)1)); if (rc) return rc;
  if (!ctx->d_vg_misc) HIPCHK(DMALLOC((void **)&ctx->d_vg_misc, 64));      // bounds[6] float, overflow flag, leaf count
  return LIVO2_OK;
}
// undistortion + voxel grid of the n > 0 raw points in d_raw / d_curv with the poses in d_poses: kernels only.  The scan-end pose comes from the host
// (rot_end / pos_end) or, with end_state, from a state on the device.  Leaves the leaves in d_xyz_aos, the overflow flag and the leaf count in d_vg_misc[8], [9].
int preprocess_enqueue(livo2_ctx *ctx, int n, int n_poses, const livo2_lidar_cfg *cfg, double leaf_size, const double *rot_end, const double *pos_end, const livo2_state *end_state) {
  HIPCHK(hipMemsetAsync(ctx->d_vg_misc, 0, 64, ctx->stream));
  HIPCHK(hipMemsetAsync(ctx->d_vg_misc, 0xFF, 12, ctx->stream));           // min codes
  if (n_poses >= 2) {
    UndistortArgs u{};
    u.xyz = ctx->d_raw; u.curvature = ctx->d_curv; u.poses = ctx->d_poses; u.n = n; u.n_poses = n_poses; u.end_state = end_state;
    // extR_Ri = Lid_rot_to_IMU^T * rot_end^T ; exrR_extT = Lid_rot_to_IMU^T * Lid_offset_to_IMU   (IMU_Processing.cpp:497-498)
    for (int i = 0; i < 3; i++) {
      if (!end_state) for (int j = 0; j < 3; j++) u.extR_Ri[i * 3 + j] = (cfg->extR[0 * 3 + i] * rot_end[j * 3 + 0] + cfg->extR[1 * 3 + i] * rot_end[j * 3 + 1]) + cfg->extR[2 * 3 + i] * rot_end[j * 3 + 2];
      u.exrR_extT[i] = (cfg->extR[0 * 3 + i] * cfg->extT[0] + cfg->extR[1 * 3 + i] * cfg->extT[1]) + cfg->extR[2 * 3 + i] * cfg->extT[2];
    }
    std::memcpy(u.ER, cfg->extR, 72); std::memcpy(u.Et, cfg->extT, 24);
    if (!end_state) std::memcpy(u.pos_end, pos_end, 24);
    hipLaunchKernelGGL(k_undistort, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, u);
  }
  uint32_t *bounds = reinterpret_cast<uint32_t *>(ctx->d_vg_misc);
  int32_t *flag = ctx->d_vg_misc + 8, *count = ctx->d_vg_misc + 9;
  const float inv_leaf = 1.0f / (float)leaf_size;
  hipLaunchKernelGGL(k_vg_minmax, dim3(std::min(64, (n + 1023) / 1024)), dim3(1024), 0, ctx->stream, ctx->d_raw, n, bounds);
  hipLaunchKernelGGL(k_vg_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_raw, n, inv_leaf, bounds, ctx->d_keys, ctx->d_idx, flag);
  size_t need = 0;
  HIPCHK(rocprim::radix_sort_pairs(nullptr, need, ctx->d_keys, ctx->d_keys2, ctx->d_idx, ctx->d_perm, (size_t)n, 0, 31, ctx->stream));
  int rc = sort_reserve(ctx, need); if (rc) return rc;
  size_t tmp_bytes = ctx->sort_tmp_bytes;
  HIPCHK(rocprim::radix_sort_pairs(ctx->d_sort_tmp, tmp_bytes, ctx->d_keys, ctx->d_keys2, ctx->d_idx, ctx->d_perm, (size_t)n, 0, 31, ctx->stream));
  hipLaunchKernelGGL(k_vg_heads, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_keys2, n, ctx->d_vg_head);
  {   // leaf number of every head = exclusive scan of the head flags (device-wide; a one-block scan cost 350 us at 240k points)
    size_t scan_need = 0;
    HIPCHK(rocprim::exclusive_scan(nullptr, scan_need, ctx->d_vg_head, ctx->d_vg_slot, 0, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
    rc = sort_reserve(ctx, scan_need); if (rc) return rc;
    size_t scan_bytes = ctx->sort_tmp_bytes;
    HIPCHK(rocprim::exclusive_scan(ctx->d_sort_tmp, scan_bytes, ctx->d_vg_head, ctx->d_vg_slot, 0, (size_t)n, rocprim::plus<int32_t>(), ctx->stream));
  }
  hipLaunchKernelGGL(k_vg_centroid, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_raw, ctx->d_keys2, ctx->d_perm, ctx->d_vg_head, ctx->d_vg_slot, n, ctx->d_xyz_aos, count);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}
} // namespace

// Raw scan -> undistortion -> voxel-grid filter -> the scan of the next update, all on the device.
int livo2_lidar_preprocess_scan(livo2_ctx *ctx, const float *xyz, const float *curvature, int32_t n, const livo2_imu_pose *poses, int32_t n_poses,
                                const double *rot_end, const double *pos_end, double leaf_size, const livo2_lidar_cfg *cfg, int32_t *n_down,
                                float *feats_undistort, float *feats_down_body) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n < 0 || (n > 0 && (!xyz || !curvature)) || n_poses < 0 || (n_poses > 0 && !poses) || !rot_end || !pos_end || !n_down) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (!(leaf_size > 0)) return fail(ctx, LIVO2_ERR_INVALID, "leaf_size must be > 0");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  for (int k = 1; k < n_poses; k++) if (poses[k].offset_time < poses[k - 1].offset_time) return fail(ctx, LIVO2_ERR_INVALID, "IMU poses must be ordered by offset_time");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  rc = preprocess_reserve(ctx, n, n_poses); if (rc) return rc;
  *n_down = 0;
  if (n == 0) { rc = scan_pipeline(ctx, 0, cfg); if (rc) return rc; ctx->has_scan = true; ctx->mt_pv_n = -1; return LIVO2_OK; }
  static_assert(sizeof(livo2_imu_pose) == 22 * 8, "livo2_imu_pose is 22 doubles");
  HIPCHK(devalloc::memcpy_async(ctx->d_raw, xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(ctx->d_curv, curvature, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n_poses > 0) HIPCHK(devalloc::memcpy_async(ctx->d_poses, poses, (size_t)n_poses * sizeof(livo2_imu_pose), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  rc = preprocess_enqueue(ctx, n, n_poses, cfg, leaf_size, rot_end, pos_end, nullptr); if (rc) return rc;
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  int32_t misc[2] = {0, 0};
  HIPCHK(devalloc::memcpy_async(misc, ctx->d_vg_misc + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (feats_undistort) HIPCHK(devalloc::memcpy_async(feats_undistort, ctx->d_raw, (size_t)n * 12, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->preprocess_kernel_us = 1e3 * ms;
  if (misc[0]) return fail(ctx, LIVO2_ERR_RANGE, "leaf size too small for the cloud: the voxel grid overflows int32 (pcl::VoxelGrid refuses it too)");
  const int m = misc[1];
  if (feats_down_body && m > 0) HIPCHK(devalloc::memcpy_async(feats_down_body, ctx->d_xyz_aos, (size_t)m * 12, hipMemcpyDeviceToHost, ctx->stream));
  rc = scan_pipeline(ctx, m, cfg); if (rc) return rc;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n_down = m;
  ctx->has_scan = true; ctx->mt_pv_n = -1;
  return LIVO2_OK;
}
double livo2_lidar_preprocess_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->preprocess_kernel_us : 0.0; }

static int lidar_ready(livo2_ctx *ctx, const livo2_state *a, const livo2_state *b, const livo2_lidar_cfg *cfg) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!a || !b) return fail(ctx, LIVO2_ERR_INVALID, "state is NULL");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  if (!ctx->has_map) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_map_upload has not been called");
  if (!ctx->has_scan) return fail(ctx, LIVO2_ERR_NO_SCAN, "livo2_lidar_set_scan has not been called");
  HIPCHK(hipSetDevice(ctx->device));
  return LIVO2_OK;
}

int livo2_lidar_iterate(livo2_ctx *ctx, const livo2_state *cur, const livo2_state *prop, const livo2_lidar_cfg *cfg, livo2_lidar_sums *sums,
                        const livo2_lidar_points *points) {
  int rc = lidar_ready(ctx, cur, prop, cfg); if (rc) return rc;
  if (!sums) return fail(ctx, LIVO2_ERR_INVALID, "sums is NULL");
  rc = ensure_lidar_outputs(ctx, points); if (rc) return rc;
  rc = upload_states(ctx, cur, prop, cfg->extR); if (rc) return rc;
  if (ctx->want_l.normal_plane) HIPCHK(hipMemsetAsync(ctx->d_normal_plane, 0xFF, (size_t)ctx->n * 4, ctx->stream));
  LidarKernelArgs a = make_lidar_args(ctx, cfg);
  const int grid = lidar_grid(std::max(ctx->n, 1), ctx->lidar_block);
  { Timed t(ctx, 0); launch_lidar_residual(ctx, a, 0); t.done(); }
  { Timed t(ctx, 2); hipLaunchKernelGGL(k_lidar_solve, dim3(1), dim3(SOLVE_THREADS), 0, ctx->stream, ctx->d_ctl, ctx->d_partials, grid, 0, 0, cfg->max_iterations, LptArgs{nullptr, nullptr, 0, 0} SOLVE_PROF_ARG); t.done(); }
  HIPCHK(hipGetLastError());
  HIPCHK(devalloc::memcpy_async(ctx->h_out, &ctx->d_ctl->sums_l, sizeof(livo2_lidar_sums), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, ctx->h_out, sizeof(livo2_lidar_sums));
  return fetch_lidar_points(ctx, points);
}

static int lidar_enqueue_loop(livo2_ctx *ctx, const livo2_lidar_cfg *cfg, int iters, int mode);
static int lidar_enqueue(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg, int iters, int mode) {
  int rc = upload_states(ctx, state_in, prop, cfg->extR); if (rc) return rc;
  return lidar_enqueue_loop(ctx, cfg, iters, mode);
}
// the iterations of one update on the states already in d_ctl
static int lidar_enqueue_loop(livo2_ctx *ctx, const livo2_lidar_cfg *cfg, int iters, int mode) {
  if (ctx->want_l.normal_plane) HIPCHK(hipMemsetAsync(ctx->d_normal_plane, 0xFF, (size_t)ctx->n * 4, ctx->stream));
  LidarKernelArgs a = make_lidar_args(ctx, cfg);
  const int grid = lidar_grid(std::max(ctx->n, 1), ctx->lidar_block);
  const LptArgs lpt = lidar_lpt_on(ctx, grid) ? LptArgs{ctx->d_lpt_cost, ctx->d_lpt_order, grid, 0} : LptArgs{nullptr, nullptr, 0, 0};
  const bool fused = lidar_fused_on(ctx);
  for (int it = 0; it < iters; it++) {
    if (fused) { Timed t(ctx, 0); launch_lidar_iteration(ctx, a, mode == 1 ? 1 : 0, mode, it % LIVO2_MAX_ITERS, mode == 1 ? iters : (1 << 30)); t.done(); continue; }
    { Timed t(ctx, 0); launch_lidar_residual(ctx, a, mode == 1 ? 1 : 0); t.done(); }
    { Timed t(ctx, 2); hipLaunchKernelGGL(k_lidar_solve, dim3(lpt.order ? 2 : 1), dim3(SOLVE_THREADS), 0, ctx->stream, ctx->d_ctl, ctx->d_partials, grid, mode, it % LIVO2_MAX_ITERS, mode == 1 ? iters : (1 << 30), lpt SOLVE_PROF_ARG); t.done(); }
    if (lpt.order) ctx->lpt_valid = true;                          // the launches enqueued from here on read the order this solve writes
  }
  if (mode != 1 || iters < 1) hipLaunchKernelGGL(k_lidar_finish, dim3(1), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->d_ctl);       // mode 1: the stopping iteration has written the result block
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

int livo2_lidar_update_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                             const livo2_lidar_points *want) {
  int rc = lidar_ready(ctx, state_in, prop, cfg); if (rc) return rc;
  rc = ensure_lidar_outputs(ctx, want); if (rc) return rc;
  return lidar_enqueue(ctx, state_in, prop, cfg, cfg->max_iterations, 1);
}

int livo2_lidar_update_fetch(livo2_ctx *ctx, livo2_lidar_result *result, const livo2_lidar_points *points) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!result) return fail(ctx, LIVO2_ERR_INVALID, "result is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(devalloc::memcpy_async(ctx->h_out, &ctx->d_ctl->lidar, sizeof(livo2_lidar_result), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memcpy(result, ctx->h_out, sizeof(livo2_lidar_result));
  int rc = fetch_lidar_points(ctx, points);
  return rc ? rc : rz_gate(ctx);
}

int livo2_lidar_update(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg, livo2_lidar_result *result,
                       const livo2_lidar_points *points) {
  int rc = livo2_lidar_update_async(ctx, state_in, prop, cfg, points); if (rc) return rc;
  return livo2_lidar_update_fetch(ctx, result, points);
}

// ---- one LiDAR-inertial frame: IMU forward propagation -> undistortion + voxel grid -> StateEstimation(state_propagat), state and scan never leave the device
int livo2_lio_frame(livo2_ctx *ctx, const livo2_state *state_in, const livo2_imu_step *steps, int32_t n_steps, const livo2_imu_cfg *imu_cfg, const livo2_imu_pose *first_pose,
                    const float *xyz, const float *curvature, int32_t n, double leaf_size, const livo2_lidar_cfg *cfg, livo2_state *state_propagat, livo2_imu_pose *poses,
                    int32_t *n_down, livo2_lidar_result *result) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!state_in || !imu_cfg || !first_pose || !n_down || !result || n_steps < 0 || n_steps > 65536 || (n_steps > 0 && !steps) || n < 0 || (n > 0 && (!xyz || !curvature)))
    return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (!(imu_cfg->mean_acc_norm > 0)) return fail(ctx, LIVO2_ERR_INVALID, "mean_acc_norm must be > 0");
  if (!(leaf_size > 0)) return fail(ctx, LIVO2_ERR_INVALID, "leaf_size must be > 0");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  if (!ctx->has_map) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_map_upload has not been called");
  for (int k = 0; k < n_steps; k++) if (steps[k].offs_t < (k ? steps[k - 1].offs_t : first_pose->offset_time)) return fail(ctx, LIVO2_ERR_INVALID, "IMU steps must be ordered by offs_t");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const int n_poses = n_steps + 1;
  if ((rc = ensure(ctx, ctx->d_imu_steps, ctx->imu_steps_cap, std::max((size_t)n_steps * 8, (size_t)8)))) return rc;
  if (!ctx->d_imu_state) HIPCHK(DMALLOC((void **)&ctx->d_imu_state, 2 * sizeof(livo2_state)));
  if ((rc = preprocess_reserve(ctx, n, n_poses))) return rc;
  *n_down = 0;
  HIPCHK(devalloc::memcpy_async(ctx->d_imu_state, state_in, sizeof(livo2_state), hipMemcpyHostToDevice, ctx->stream));
  if (n_steps > 0) HIPCHK(devalloc::memcpy_async(ctx->d_imu_steps, steps, (size_t)n_steps * 64, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(ctx->d_poses, first_pose, sizeof(livo2_imu_pose), hipMemcpyHostToDevice, ctx->stream));      // IMUpose[0] (IMU_Processing.cpp:312-313)
  if (n > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_raw, xyz, (size_t)n * 12, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_curv, curvature, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  // 1. forward propagation: d_imu_state[1] = state_propagat, IMUpose[1..] behind the first pose
  ImuKernelArgs ia = make_imu_args(ctx, imu_cfg, n_steps, ctx->d_poses + 22);
  hipLaunchKernelGGL(k_imu_propagate, dim3(1), dim3(IMU_THREADS), 0, ctx->stream, ia);
  HIPCHK(hipGetLastError());
  // 2. backward propagation to the scan-end pose (read from the device state) + voxel grid
  if (n > 0) { rc = preprocess_enqueue(ctx, n, n_poses, cfg, leaf_size, nullptr, nullptr, ctx->d_imu_state + 1); if (rc) return rc; }
  int32_t misc[2] = {0, 0};
  if (n > 0) HIPCHK(devalloc::memcpy_async(misc, ctx->d_vg_misc + 8, 8, hipMemcpyDeviceToHost, ctx->stream));
  if (state_propagat) HIPCHK(devalloc::memcpy_async(state_propagat, ctx->d_imu_state + 1, sizeof(livo2_state), hipMemcpyDeviceToHost, ctx->stream));
  if (poses && n_steps > 0) HIPCHK(devalloc::memcpy_async(poses, ctx->d_poses + 22, (size_t)n_steps * sizeof(livo2_imu_pose), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));        // the leaf count sizes the launches of the update
  if (misc[0]) return fail(ctx, LIVO2_ERR_RANGE, "leaf size too small for the cloud: the voxel grid overflows int32 (pcl::VoxelGrid refuses it too)");
  const int m = misc[1];
  if ((rc = scan_pipeline(ctx, m, cfg))) return rc;
  ctx->has_scan = true; ctx->mt_pv_n = -1;
  *n_down = m;
  // 3. StateEstimation(state_propagat) with state_ = state_propagat (LIVMapper.cpp:366-370)
  if ((rc = ensure_lidar_outputs(ctx, nullptr))) return rc;
  Mat9 er; std::memcpy(er.v, cfg->extR, 72);
  hipLaunchKernelGGL(k_ctl_from_state, dim3(1), dim3(256), 0, ctx->stream, ctx->d_ctl, ctx->d_imu_state + 1, er);
  if ((rc = lidar_enqueue_loop(ctx, cfg, cfg->max_iterations, 1))) return rc;
  return livo2_lidar_update_fetch(ctx, result, nullptr);
}

int livo2_lidar_iterations_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg, int32_t iters) {
  int rc = lidar_ready(ctx, state_in, prop, cfg); if (rc) return rc;
  if (iters < 1) return fail(ctx, LIVO2_ERR_INVALID, "iters must be >= 1");
  rc = ensure_lidar_outputs(ctx, nullptr); if (rc) return rc;
  return lidar_enqueue(ctx, state_in, prop, cfg, iters, 2);
}

// ---- batch of frames ---------------------------------------------------------------------------------------------------------
// B independent StateEstimation problems (own scan, own states) against the resident map.  Every ESIKF iteration is ONE residual
// grid over all frames plus one solve block per frame; a frame that has stopped (hdr.stop) drops out of later grids at its
// blocks' first instruction.  Same per-point arithmetic and decisions as B separate livo2_lidar_update calls; the partial sums are grouped in
// LIDAR_BLOCK_BATCH-point blocks here (LIDAR_BLOCK there), so the results agree to rounding, not to the last bit.
namespace {

__global__ void __launch_bounds__(LIVO2_WAVE) k_batch_scatter_in(const HostIn *__restrict__ in, DevCtl *__restrict__ ctl) {
  const double *src = reinterpret_cast<const double *>(in + blockIdx.x);
  double *dst = reinterpret_cast<double *>(ctl + blockIdx.x);
  for (int e = threadIdx.x; e < (int)(sizeof(HostIn) / sizeof(double)); e += LIVO2_WAVE) dst[e] = src[e];
}
__global__ void __launch_bounds__(LIVO2_WAVE) k_batch_gather_out(const DevCtl *__restrict__ ctl, livo2_lidar_result *__restrict__ out) {
  const double *src = reinterpret_cast<const double *>(&ctl[blockIdx.x].lidar);
  double *dst = reinterpret_cast<double *>(out + blockIdx.x);
  for (int e = threadIdx.x; e < (int)(sizeof(livo2_lidar_result) / sizeof(double)); e += LIVO2_WAVE) dst[e] = src[e];
}
static_assert(sizeof(HostIn) % 8 == 0 && sizeof(livo2_lidar_result) % 8 == 0 && sizeof(DevCtl) % 8 == 0, "copied as doubles");

int batch_alloc_fixed(livo2_ctx *ctx) {
  if (ctx->bd_ctl) return LIVO2_OK;
  HIPCHK(DMALLOC((void **)&ctx->bd_ctl, sizeof(DevCtl) * LIVO2_MAX_BATCH));
  HIPCHK(DMALLOC((void **)&ctx->bd_entries, sizeof(LidarBatchEntry) * LIVO2_MAX_BATCH));
  HIPCHK(DMALLOC((void **)&ctx->bd_in, sizeof(HostIn) * LIVO2_MAX_BATCH));
  HIPCHK(DMALLOC((void **)&ctx->bd_results, sizeof(livo2_lidar_result) * LIVO2_MAX_BATCH));
  HIPCHK(hipHostMalloc((void **)&ctx->bh_in, sizeof(HostIn) * LIVO2_MAX_BATCH));
  HIPCHK(hipHostMalloc((void **)&ctx->bh_results, sizeof(livo2_lidar_result) * LIVO2_MAX_BATCH));
  HIPCHK(hipHostMalloc((void **)&ctx->bh_entries, sizeof(LidarBatchEntry) * LIVO2_MAX_BATCH));
  HIPCHK(hipMemsetAsync(ctx->bd_ctl, 0, sizeof(DevCtl) * LIVO2_MAX_BATCH, ctx->stream));
  return LIVO2_OK;
}

} // namespace

int livo2_lidar_batch_set_scans(livo2_ctx *ctx, int32_t n_frames, const float *xyz, const int32_t *counts, const livo2_lidar_cfg *cfg) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n_frames < 1 || n_frames > LIVO2_MAX_BATCH) return fail(ctx, LIVO2_ERR_INVALID, "n_frames out of [1,LIVO2_MAX_BATCH]");
  if (!counts) return fail(ctx, LIVO2_ERR_INVALID, "counts is NULL");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  long long total = 0;
  for (int f = 0; f < n_frames; f++) { if (counts[f] < 0) return fail(ctx, LIVO2_ERR_INVALID, "negative point count"); total += counts[f]; }
  if (total > 0 && !xyz) return fail(ctx, LIVO2_ERR_INVALID, "xyz is NULL");
  if (total > (1ll << 30)) return fail(ctx, LIVO2_ERR_INVALID, "batch too large");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  rc = batch_alloc_fixed(ctx); if (rc) return rc;
  if ((int)total > ctx->b_cap) {
    hipError_t e;
    if (ctx->bd_x) { e = DFREE(ctx->bd_xyz_aos); e = DFREE(ctx->bd_x); e = DFREE(ctx->bd_y); e = DFREE(ctx->bd_z); e = DFREE(ctx->bd_cb); e = DFREE(ctx->bd_keys); e = DFREE(ctx->bd_keys2); e = DFREE(ctx->bd_idx); e = DFREE(ctx->bd_perm); (void)e; }
    ctx->bd_x = nullptr;
    size_t cap = std::max((size_t)total, (size_t)1024);
    HIPCHK(DMALLOC((void **)&ctx->bd_xyz_aos, cap * 12)); HIPCHK(DMALLOC((void **)&ctx->bd_x, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->bd_y, cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->bd_z, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->bd_cb, cap * 48)); HIPCHK(DMALLOC((void **)&ctx->bd_keys, cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->bd_keys2, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->bd_idx, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->bd_perm, cap * 4));
    ctx->b_cap = (int)cap;
  }
  ctx->bn = n_frames; ctx->b_total = (int)total;
  ctx->b_count.assign(counts, counts + n_frames);
  ctx->b_off.resize(n_frames); ctx->b_grid.resize(n_frames); ctx->b_block_begin.resize(n_frames);
  int off = 0, blocks = 0;
  for (int f = 0; f < n_frames; f++) {
    ctx->b_off[f] = off; ctx->b_grid[f] = lidar_grid(std::max(counts[f], 1), LIDAR_BLOCK_BATCH); ctx->b_block_begin[f] = blocks;
    off += counts[f]; blocks += ctx->b_grid[f];
  }
  ctx->b_blocks = blocks;
  rc = ensure(ctx, ctx->bd_partials, ctx->b_partials_cap, (size_t)blocks * 32); if (rc) return rc;
  rc = ensure(ctx, ctx->bd_block_frame, ctx->b_block_frame_cap, (size_t)blocks); if (rc) return rc;
  {
    std::vector<int32_t> bf((size_t)blocks);
    for (int f = 0; f < n_frames; f++) std::fill(bf.begin() + ctx->b_block_begin[f], bf.begin() + ctx->b_block_begin[f] + ctx->b_grid[f], f);
    HIPCHK(hipMemcpy(ctx->bd_block_frame, bf.data(), (size_t)blocks * 4, hipMemcpyHostToDevice));
  }
  if (total > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->bd_xyz_aos, xyz, (size_t)total * 12, hipMemcpyHostToDevice, ctx->stream));
    const double deg2rad = cfg->deg2rad != 0.0 ? cfg->deg2rad : 0.017453293;
    int nmax = 0; for (int f = 0; f < n_frames; f++) nmax = std::max(nmax, counts[f]);
    size_t need = 0;
    HIPCHK(rocprim::radix_sort_pairs(nullptr, need, ctx->bd_keys, ctx->bd_keys2, ctx->bd_idx, ctx->bd_perm, (size_t)nmax, 0, 30, ctx->stream));
    if (need > ctx->sort_tmp_bytes) {
      HIPCHK(hipStreamSynchronize(ctx->stream));
      if (ctx->d_sort_tmp) HIPCHK(DFREE(ctx->d_sort_tmp));
      ctx->d_sort_tmp = nullptr;
      HIPCHK(DMALLOC(&ctx->d_sort_tmp, need + need / 2 + 256));
      ctx->sort_tmp_bytes = need + need / 2 + 256;
    }
    for (int f = 0; f < n_frames; f++) {          // same per-scan pipeline as livo2_lidar_set_scan, on this frame's slice
      const int n = counts[f], o = ctx->b_off[f];
      if (n == 0) continue;
      hipLaunchKernelGGL(k_morton_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->bd_xyz_aos + (size_t)o * 3, n, (float)(1.0 / cfg->voxel_size), ctx->bd_keys + o, ctx->bd_idx + o);
      size_t tmp_bytes = ctx->sort_tmp_bytes;
      HIPCHK(rocprim::radix_sort_pairs(ctx->d_sort_tmp, tmp_bytes, ctx->bd_keys + o, ctx->bd_keys2 + o, ctx->bd_idx + o, ctx->bd_perm + o, (size_t)n, 0, 30, ctx->stream));
      hipLaunchKernelGGL(k_gather_xyz, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->bd_xyz_aos + (size_t)o * 3, ctx->bd_perm + o, n, ctx->bd_x + o, ctx->bd_y + o, ctx->bd_z + o);
      hipLaunchKernelGGL(k_body_cov, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->bd_x + o, ctx->bd_y + o, ctx->bd_z + o, n, (float)cfg->dept_err, (float)cfg->beam_err,
                         deg2rad, ctx->bd_cb + (size_t)o * 6);
    }
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->has_batch = true;
  return LIVO2_OK;
}

static int batch_enqueue(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg, int iters, int mode) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!state_in || !prop) return fail(ctx, LIVO2_ERR_INVALID, "state is NULL");
  int rc = check_lidar_cfg(ctx, cfg); if (rc) return rc;
  if (!ctx->has_map) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_map_upload has not been called");
  if (!ctx->has_batch) return fail(ctx, LIVO2_ERR_NO_SCAN, "livo2_lidar_batch_set_scans has not been called");
  if (n_frames != ctx->bn) return fail(ctx, LIVO2_ERR_INVALID, "n_frames differs from the batch set by livo2_lidar_batch_set_scans");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));        // pinned staging blocks are reused
  for (int f = 0; f < n_frames; f++) {
    HostIn &h = ctx->bh_in[f];
    h.cur = state_in[f]; h.prop = prop[f];
    std::memset(&h.hdr, 0, sizeof(DevHeader));
    h.hdr.last_error = FLT_MAX;
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
      h.hdr.RE[i * 3 + j] = (prop[f].rot[i * 3] * cfg->extR[j] + prop[f].rot[i * 3 + 1] * cfg->extR[3 + j]) + prop[f].rot[i * 3 + 2] * cfg->extR[6 + j];
    LidarBatchEntry &e = ctx->bh_entries[f];
    std::memset(&e, 0, sizeof(e));
    const int o = ctx->b_off[f];
    e.a.x = ctx->bd_x + o; e.a.y = ctx->bd_y + o; e.a.z = ctx->bd_z + o; e.a.cb = ctx->bd_cb + (size_t)o * 6; e.a.perm = ctx->bd_perm + o;
    e.a.n = ctx->b_count[f]; e.a.max_layer = cfg->max_layer; e.a.map = ctx->map; e.a.voxel_size = cfg->voxel_size; e.a.sigma_num = cfg->sigma_num;
    std::memcpy(e.a.ER, cfg->extR, 72); std::memcpy(e.a.Et, cfg->extT, 24);
    e.ctl = ctx->bd_ctl + f; e.partials = ctx->bd_partials + (size_t)ctx->b_block_begin[f] * 32; e.block_begin = ctx->b_block_begin[f]; e.nblocks = ctx->b_grid[f];
  }
  HIPCHK(devalloc::memcpy_async(ctx->bd_in, ctx->bh_in, sizeof(HostIn) * n_frames, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(ctx->bd_entries, ctx->bh_entries, sizeof(LidarBatchEntry) * n_frames, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_batch_scatter_in, dim3(n_frames), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->bd_in, ctx->bd_ctl);
  for (int it = 0; it < iters; it++) {
    { Timed t(ctx, 0); hipLaunchKernelGGL(k_lidar_residual_batch, dim3(ctx->b_blocks), dim3(LIDAR_BLOCK_BATCH), LIDAR_LDS_BYTES_OF(LIDAR_BLOCK_BATCH) + LIDAR_LDS_DUMP, ctx->stream, ctx->bd_entries, ctx->bd_block_frame, mode == 1 ? 1 : 0); t.done(); }
    { Timed t(ctx, 2); hipLaunchKernelGGL(k_lidar_solve_batch, dim3(n_frames), dim3(SOLVE_THREADS), 0, ctx->stream, ctx->bd_entries, mode, it % LIVO2_MAX_ITERS, mode == 1 ? iters : (1 << 30)); t.done(); }
  }
  if (mode != 1 || iters < 1) hipLaunchKernelGGL(k_lidar_finish, dim3(n_frames), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->bd_ctl);
  hipLaunchKernelGGL(k_batch_gather_out, dim3(n_frames), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->bd_ctl, ctx->bd_results);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

int livo2_lidar_batch_update_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg) {
  if (!ctx || !cfg) return ctx ? fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL") : LIVO2_ERR_INVALID;
  return batch_enqueue(ctx, n_frames, state_in, prop, cfg, cfg->max_iterations, 1);
}

int livo2_lidar_batch_update_fetch(livo2_ctx *ctx, int32_t n_frames, livo2_lidar_result *results) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!results) return fail(ctx, LIVO2_ERR_INVALID, "results is NULL");
  if (!ctx->has_batch || n_frames != ctx->bn) return fail(ctx, LIVO2_ERR_INVALID, "n_frames differs from the batch set by livo2_lidar_batch_set_scans");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(devalloc::memcpy_async(ctx->bh_results, ctx->bd_results, sizeof(livo2_lidar_result) * n_frames, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memcpy(results, ctx->bh_results, sizeof(livo2_lidar_result) * n_frames);
  return rz_gate(ctx);
}

int livo2_lidar_batch_update(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                             livo2_lidar_result *results) {
  int rc = livo2_lidar_batch_update_async(ctx, n_frames, state_in, prop, cfg); if (rc) return rc;
  return livo2_lidar_batch_update_fetch(ctx, n_frames, results);
}

int livo2_lidar_batch_iterations_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_lidar_cfg *cfg,
                                       int32_t iters) {
  if (!ctx || !cfg) return ctx ? fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL") : LIVO2_ERR_INVALID;
  if (iters < 1) return fail(ctx, LIVO2_ERR_INVALID, "iters must be >= 1");
  return batch_enqueue(ctx, n_frames, state_in, prop, cfg, iters, 2);
}

// ---- visual ------------------------------------------------------------------------------------------------------------------
int livo2_visual_set_frame(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const double *pos, const float *warp_patch,
                           const int32_t *search_levels, const double *inv_expo_list, int32_t M, int32_t L) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!img || width <= 0 || height <= 0 || stride < width) return fail(ctx, LIVO2_ERR_INVALID, "bad image");
  if (M < 0 || L < 1 || L > LIVO2_MAX_LEVELS || (M > 0 && (!pos || !warp_patch || !search_levels || !inv_expo_list))) return fail(ctx, LIVO2_ERR_INVALID, "bad sub-map arrays");
  for (int i = 0; i < M; i++) if (search_levels[i] < 0 || search_levels[i] > 8) return fail(ctx, LIVO2_ERR_RANGE, "search_level out of [0,8]");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc = ensure(ctx, ctx->d_img, ctx->img_cap, (size_t)stride * height); if (rc) return rc;
  if (M > ctx->M_cap) {
    hipError_t e;
    if (ctx->d_pos) { e = DFREE(ctx->d_pos); e = DFREE(ctx->d_invexpo); e = DFREE(ctx->d_search); e = DFREE(ctx->d_errors); (void)e; }
    int cap = std::max(M, 512);
    HIPCHK(DMALLOC((void **)&ctx->d_pos, (size_t)cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_invexpo, (size_t)cap * 8));
    HIPCHK(DMALLOC((void **)&ctx->d_search, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_errors, (size_t)cap * 4));
    ctx->M_cap = cap;
  }
  rc = ensure(ctx, ctx->d_warp, ctx->warp_cap, std::max((size_t)M * L * 64, (size_t)64)); if (rc) return rc;
  const int grid = visual_grid_inverse(std::max(M, 1));     // the larger of the two grids
  rc = ensure(ctx, ctx->d_partials, ctx->partials_cap, std::max((size_t)grid * VIS_PSTRIDE, (size_t)64)); if (rc) return rc;
  HIPCHK(devalloc::memcpy_async(ctx->d_img, img, (size_t)stride * height, hipMemcpyHostToDevice, ctx->stream));
  if (M > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_pos, pos, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_warp, warp_patch, (size_t)M * L * 256, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_search, search_levels, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_invexpo, inv_expo_list, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->width = width; ctx->height = height; ctx->stride = stride; ctx->M = M; ctx->L = L;
  ctx->has_frame = true;
  ctx->has_ref = false;
  return LIVO2_OK;
}

int livo2_visual_set_reference(livo2_ctx *ctx, const uint8_t *ref_imgs, int32_t n_ref, const int32_t *ref_img_idx, const double *ref_px, const double *ref_f,
                               const double *ref_R, const double *ref_pos) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!ctx->has_frame) return fail(ctx, LIVO2_ERR_NO_FRAME, "livo2_visual_set_frame has not been called");
  const int M = ctx->M;
  if (n_ref < 1 || !ref_imgs || (M > 0 && (!ref_img_idx || !ref_px || !ref_f || !ref_R || !ref_pos))) return fail(ctx, LIVO2_ERR_INVALID, "bad reference arrays");
  for (int i = 0; i < M; i++) if (ref_img_idx[i] < 0 || ref_img_idx[i] >= n_ref) return fail(ctx, LIVO2_ERR_INVALID, "ref_img_idx out of range");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t img_bytes = (size_t)ctx->stride * ctx->height;
  int rc = ensure(ctx, ctx->d_ref_imgs, ctx->ref_img_cap, img_bytes * n_ref); if (rc) return rc;
  if (M > ctx->ref_cap) {
    hipError_t e;
    if (ctx->d_ref_idx) { e = DFREE(ctx->d_ref_idx); e = DFREE(ctx->d_ref_px); e = DFREE(ctx->d_ref_f); e = DFREE(ctx->d_ref_R); e = DFREE(ctx->d_ref_pos); e = DFREE(ctx->d_gref); e = DFREE(ctx->d_mref); (void)e; }
    const size_t cap = (size_t)std::max(M, 512);
    HIPCHK(DMALLOC((void **)&ctx->d_ref_idx, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_ref_px, cap * 16)); HIPCHK(DMALLOC((void **)&ctx->d_ref_f, cap * 24));
    HIPCHK(DMALLOC((void **)&ctx->d_ref_R, cap * 72)); HIPCHK(DMALLOC((void **)&ctx->d_ref_pos, cap * 24));
    HIPCHK(DMALLOC((void **)&ctx->d_gref, cap * 64 * 16)); HIPCHK(DMALLOC((void **)&ctx->d_mref, cap * 16 * 8));
    ctx->ref_cap = (int)cap;
  }
  HIPCHK(devalloc::memcpy_async(ctx->d_ref_imgs, ref_imgs, img_bytes * n_ref, hipMemcpyHostToDevice, ctx->stream));
  if (M > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_idx, ref_img_idx, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_px, ref_px, (size_t)M * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_f, ref_f, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_R, ref_R, (size_t)M * 72, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_pos, ref_pos, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_ref = n_ref;
  ctx->has_ref = true;
  return LIVO2_OK;
}

// ---- visual sub-map retrieval, selection half -----------------------------------------------------------------------------------
int livo2_visual_map_upload(livo2_ctx *ctx, int32_t n, const double *pos, const int64_t *voxel_key, const uint8_t *active) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n < 0 || (n > 0 && !pos)) return fail(ctx, LIVO2_ERR_INVALID, "bad visual map arrays");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc;
  if ((rc = ensure(ctx, ctx->d_vm_pos, ctx->vm_pos_cap, std::max((size_t)n * 3, (size_t)3)))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_pkey, ctx->vm_pkey_cap, std::max((size_t)n, (size_t)1)))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_active, ctx->vm_active_cap, std::max((size_t)n, (size_t)1)))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_fov, ctx->vm_fov_cap, std::max((size_t)n, (size_t)1)))) return rc;
  if (!ctx->d_sel_flag) HIPCHK(DMALLOC((void **)&ctx->d_sel_flag, 64));
  HIPCHK(hipMemsetAsync(ctx->d_sel_flag, 0, 64, ctx->stream));
  if (n > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_vm_pos, pos, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    if (active) HIPCHK(devalloc::memcpy_async(ctx->d_vm_active, active, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    else HIPCHK(hipMemsetAsync(ctx->d_vm_active, 1, (size_t)n, ctx->stream));
    if (voxel_key) {
      std::vector<unsigned long long> pk((size_t)n);
      const long long B = 1ll << 20;
      for (int i = 0; i < n; i++) {
        const int64_t *k = voxel_key + 3 * (size_t)i;
        if (k[0] < -B || k[0] >= B || k[1] < -B || k[1] >= B || k[2] < -B || k[2] >= B) return fail(ctx, LIVO2_ERR_RANGE, "visual voxel key outside 21 bits per axis");
        pk[i] = ((unsigned long long)(k[0] + B) << 42) | ((unsigned long long)(k[1] + B) << 21) | (unsigned long long)(k[2] + B);
      }
      HIPCHK(hipMemcpy(ctx->d_vm_pkey, pk.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    } else {
      hipLaunchKernelGGL(k_sel_point_keys, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_vm_pos, n, ctx->d_vm_pkey, ctx->d_sel_flag);
      HIPCHK(hipGetLastError());
    }
  }
  {                                                    // the voxels that hold visual points, as a set (the RayCasting module asks "is there anything in this voxel")
    size_t cap = 1024; while (cap < 2 * (size_t)std::max(n, 1)) cap <<= 1;
    if ((rc = ensure(ctx, ctx->d_vm_set, ctx->vm_set_cap, cap))) return rc;
    ctx->vm_set_mask = (uint32_t)(cap - 1);
    HIPCHK(hipMemsetAsync(ctx->d_vm_set, 0xFF, cap * 8, ctx->stream));
    if (n > 0) hipLaunchKernelGGL(k_vm_voxel_set, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_vm_pkey, n, ctx->d_vm_set, ctx->vm_set_mask);
    HIPCHK(hipGetLastError());
  }
  int32_t flag = 0;
  HIPCHK(devalloc::memcpy_async(&flag, ctx->d_sel_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (flag) return fail(ctx, LIVO2_ERR_RANGE, "visual voxel key outside 21 bits per axis");
  ctx->n_vm = n; ctx->has_vmap = true;
  ctx->has_obs = false;                       // the observation table belongs to the previous point set
  return LIVO2_OK;
}

// selection stage: buffers for n_pg scan points and `length` grid cells; *cap_out = capacity of the scan-voxel hash set
static int select_reserve(livo2_ctx *ctx, const livo2_select_cfg *cfg, int32_t n_pg, size_t *cap_out) {
  if (!ctx->has_vmap) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_visual_map_upload has not been called");
  const int length = cfg->grid_n_width * cfg->grid_n_height;
  if (cfg->raycast_en != 0 && cfg->raycast_en != 1) return fail(ctx, LIVO2_ERR_INVALID, "raycast_en must be 0 or 1 (the field was padding before livo2_hip 0.2: zero-initialise livo2_select_cfg)");
  if (cfg->grid_size < 1 || cfg->grid_n_width < 1 || cfg->grid_n_height < 1 || length > (1 << 20) || cfg->cam.width < 1 || cfg->cam.height < 1 || cfg->patch_size_half < 0 ||
      cfg->border < cfg->patch_size_half) return fail(ctx, LIVO2_ERR_INVALID, "bad grid / border (the 9x9 depth window must stay inside the image: border >= patch_size_half)");
  const size_t px = (size_t)cfg->cam.width * cfg->cam.height;
  size_t cap = 1024; while (cap < 2 * (size_t)std::max(n_pg, 1)) cap <<= 1;
  int rc;
  if ((rc = ensure(ctx, ctx->d_sel_pg, ctx->sel_pg_cap, std::max((size_t)n_pg * 3, (size_t)3)))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_set, ctx->sel_set_cap, cap))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_depth, ctx->sel_depth_cap, px))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_best, ctx->sel_best_cap, (size_t)length))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_type, ctx->sel_type_cap, (size_t)length))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_point, ctx->sel_point_cap, (size_t)length))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_dist, ctx->sel_dist_cap, (size_t)length))) return rc;
  if ((rc = ensure(ctx, ctx->d_sel_disc, ctx->sel_disc_cap, (size_t)length))) return rc;
  if (cfg->raycast_en) {
    if (length > RAY_MAX_CELLS) return fail(ctx, LIVO2_ERR_INVALID, "raycast_en: more than RAY_MAX_CELLS (32768) grid cells");
    if (ctx->has_map && !ctx->tree_mode) return fail(ctx, LIVO2_ERR_NO_MAP, "raycast_en looks into the LiDAR VoxelMap (plane_map, vio.cpp:573-585): it needs the device-resident map (livo2_map_tree_*), not a snapshot");
    size_t rcap = 1024; while (rcap < 2 * (size_t)length) rcap <<= 1;
    const size_t nv = (size_t)std::max(ctx->n_vm, 1);
    if ((rc = ensure(ctx, ctx->d_ray_set, ctx->ray_set_cap, rcap))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_key, ctx->ray_key_cap, (size_t)length))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_action, ctx->ray_action_cap, (size_t)length))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_hit_key, ctx->ray_hit_key_cap, nv))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_hit_best, ctx->ray_hit_best_cap, nv))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_hit_cell, ctx->ray_hit_cell_cap, nv))) return rc;
    if ((rc = ensure(ctx, ctx->d_ray_add, ctx->ray_add_cap, (size_t)length * 6))) return rc;
    if (!ctx->d_ray_counters) HIPCHK(DMALLOC((void **)&ctx->d_ray_counters, 64));
  }
  *cap_out = cap;
  return LIVO2_OK;
}

// new_frame_->pos() = -(R^T t)
static void frame_pos(const double *R, const double *t, double *o) { for (int r = 0; r < 3; r++) o[r] = ((R[r] * t[0] + R[3 + r] * t[1]) + R[6 + r] * t[2]) * (-1.0); }

// selection stage: resets + the three kernels on the ctx stream (the scan points are already in d_sel_pg)
static int select_enqueue(livo2_ctx *ctx, const livo2_select_cfg *cfg, int32_t n_pg, size_t cap) {
  const int length = cfg->grid_n_width * cfg->grid_n_height;
  const size_t px = (size_t)cfg->cam.width * cfg->cam.height;
  SelectArgs a{};
  a.fx = cfg->cam.fx; a.fy = cfg->cam.fy; a.cx = cfg->cam.cx; a.cy = cfg->cam.cy; std::memcpy(a.d, cfg->cam.d, 40); a.distortion = cfg->cam.distortion;
  std::memcpy(a.R, cfg->R_cur, 72); std::memcpy(a.t, cfg->t_cur, 24);
  frame_pos(cfg->R_cur, cfg->t_cur, a.cam_pos);
  a.width = cfg->cam.width; a.height = cfg->cam.height; a.border = cfg->border; a.grid_size = cfg->grid_size; a.grid_n_width = cfg->grid_n_width; a.length = length;
  a.patch_size_half = cfg->patch_size_half; a.n_pg = n_pg; a.n_pts = ctx->n_vm;
  a.pg = ctx->d_sel_pg; a.pos = ctx->d_vm_pos; a.pkey = ctx->d_vm_pkey; a.active = ctx->d_vm_active; a.set = ctx->d_sel_set; a.mask = (uint32_t)(cap - 1);
  a.depth = ctx->d_sel_depth; a.cell_best = ctx->d_sel_best; a.cell_type = ctx->d_sel_type; a.in_fov = ctx->d_vm_fov; a.range_flag = ctx->d_sel_flag;
  a.cell_point = ctx->d_sel_point; a.cell_dist = ctx->d_sel_dist; a.cell_discont = ctx->d_sel_disc;
  {
    const size_t most = std::max(std::max(cap, px), std::max((size_t)length, (size_t)16));
    hipLaunchKernelGGL(k_sel_reset, dim3((unsigned)((most + 255) / 256)), dim3(256), 0, ctx->stream, a, (uint32_t)cap, (uint32_t)px);
  }
  if (n_pg > 0) hipLaunchKernelGGL(k_sel_scan, dim3((n_pg + 255) / 256), dim3(256), 0, ctx->stream, a);
  if (ctx->n_vm > 0) hipLaunchKernelGGL(k_sel_points, dim3((ctx->n_vm + 255) / 256), dim3(256), 0, ctx->stream, a);
  ctx->ray_add_n = -1;
  if (cfg->raycast_en) {                               // RayCasting module (vio.cpp:487-591) between stages B and C
    RayArgs r{};
    r.s = a;
    r.vmset = ctx->d_vm_set; r.vm_mask = ctx->vm_set_mask;
    size_t rcap = 1024; while (rcap < 2 * (size_t)length) rcap <<= 1;
    r.rayset = ctx->d_ray_set; r.ray_mask = (uint32_t)(rcap - 1);
    r.action = ctx->d_ray_action; r.key = ctx->d_ray_key; r.hit_key = ctx->d_ray_hit_key; r.hit_best = ctx->d_ray_hit_best; r.hit_cell = ctx->d_ray_hit_cell;
    r.counters = ctx->d_ray_counters; r.add6 = ctx->d_ray_add;
    if (ctx->tree_mode) {
      r.nodes = ctx->mt.nodes; r.slots = ctx->d_slots; r.planes = ctx->d_planes; r.lmask = ctx->mt.mask; r.lseed1 = ctx->mt.seed1; r.lseed2 = ctx->mt.seed2; r.max_layer = ctx->mt.max_layer;
    }
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.Rt[i * 3 + j] = cfg->R_cur[j * 3 + i];
    for (int i = 0; i < 3; i++) r.tinv[i] = ((r.Rt[i * 3] * cfg->t_cur[0] + r.Rt[i * 3 + 1] * cfg->t_cur[1]) + r.Rt[i * 3 + 2] * cfg->t_cur[2]) * (-1.0);
    HIPCHK(hipMemsetAsync(ctx->d_ray_set, 0xFF, rcap * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->d_ray_counters, 0, 64, ctx->stream));
    hipLaunchKernelGGL(k_ray_find, dim3((length + 255) / 256), dim3(256), 0, ctx->stream, r);
    if (ctx->n_vm > 0) hipLaunchKernelGGL(k_ray_points, dim3((ctx->n_vm + 255) / 256), dim3(256), 0, ctx->stream, r);
    hipLaunchKernelGGL(k_ray_resolve, dim3(1), dim3(256), 0, ctx->stream, r);
    ctx->ray_add_n = 0;                                 // (the count stays on the device until livo2_visual_raycast_fetch)
  }
  hipLaunchKernelGGL(k_sel_cells, dim3((length + 3) / 4), dim3(256), 0, ctx->stream, a);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

int livo2_visual_select(livo2_ctx *ctx, const double *pg, int32_t n_pg, const livo2_select_cfg *cfg, int32_t *cell_point, float *cell_dist, uint8_t *cell_disc,
                        uint8_t *point_in_fov) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!cfg || n_pg < 0 || (n_pg > 0 && !pg) || !cell_point) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  size_t cap = 0;
  int rc = select_reserve(ctx, cfg, n_pg, &cap); if (rc) return rc;
  const int length = cfg->grid_n_width * cfg->grid_n_height;
  if (n_pg > 0) HIPCHK(devalloc::memcpy_async(ctx->d_sel_pg, pg, (size_t)n_pg * 24, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  if ((rc = select_enqueue(ctx, cfg, n_pg, cap))) return rc;
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  int32_t flag = 0;
  HIPCHK(devalloc::memcpy_async(&flag, ctx->d_sel_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(cell_point, ctx->d_sel_point, (size_t)length * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (cell_dist) HIPCHK(devalloc::memcpy_async(cell_dist, ctx->d_sel_dist, (size_t)length * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (cell_disc) HIPCHK(devalloc::memcpy_async(cell_disc, ctx->d_sel_disc, (size_t)length, hipMemcpyDeviceToHost, ctx->stream));
  if (point_in_fov && ctx->n_vm > 0) HIPCHK(devalloc::memcpy_async(point_in_fov, ctx->d_vm_fov, (size_t)ctx->n_vm, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->select_kernel_us = 1e3 * ms;
  if (flag) return fail(ctx, LIVO2_ERR_RANGE, "scan voxel key outside 21 bits per axis");
  return LIVO2_OK;
}
double livo2_visual_select_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->select_kernel_us : 0.0; }

// ---- visual sub-map retrieval, per-point tail -----------------------------------------------------------------------------------
// tail stage: candidate arrays, per-candidate outputs and the frame arrays for up to n candidates
static int tail_reserve(livo2_ctx *ctx, int n, int L) {
  if (n > ctx->cand_cap) {
    void *old[] = {ctx->d_c_pos, ctx->d_c_normal, ctx->d_c_px, ctx->d_c_f, ctx->d_c_R, ctx->d_c_t, ctx->d_c_ie, ctx->d_c_ncc, ctx->d_c_A, ctx->d_c_idx, ctx->d_c_lvl,
                   ctx->d_c_acc, ctx->d_c_sl, ctx->d_c_slot, ctx->d_c_err};
    for (void *p : old) if (p) { hipError_t e = DFREE(p); (void)e; }
    const size_t cap = (size_t)std::max(n, 512);
    HIPCHK(DMALLOC((void **)&ctx->d_c_pos, cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_c_normal, cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_c_px, cap * 16));
    HIPCHK(DMALLOC((void **)&ctx->d_c_f, cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_c_R, cap * 72)); HIPCHK(DMALLOC((void **)&ctx->d_c_t, cap * 24));
    HIPCHK(DMALLOC((void **)&ctx->d_c_ie, cap * 8)); HIPCHK(DMALLOC((void **)&ctx->d_c_ncc, cap * 8)); HIPCHK(DMALLOC((void **)&ctx->d_c_A, cap * 32));
    HIPCHK(DMALLOC((void **)&ctx->d_c_idx, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_c_lvl, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_c_acc, cap * 4));
    HIPCHK(DMALLOC((void **)&ctx->d_c_sl, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_c_slot, cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_c_err, cap * 4));
    ctx->cand_cap = (int)cap;
  }
  if (!ctx->d_c_count) HIPCHK(DMALLOC((void **)&ctx->d_c_count, 64));
  int rc = ensure(ctx, ctx->d_c_patch, ctx->c_patch_cap, std::max((size_t)n * L * 64, (size_t)64)); if (rc) return rc;
  size_t ld = 1024; while (ld < 2 * (size_t)std::max(n, 1)) ld <<= 1;
  if ((rc = ensure(ctx, ctx->d_c_id, ctx->c_id_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->d_c_leader, ctx->c_leader_cap, (size_t)std::max(n, 1)))) return rc;
  if ((rc = ensure(ctx, ctx->d_ld_keys, ctx->ld_keys_cap, ld))) return rc;
  if ((rc = ensure(ctx, ctx->d_ld_vals, ctx->ld_vals_cap, ld))) return rc;
  // the frame arrays must be able to hold every candidate
  if (n > ctx->M_cap) {
    hipError_t e;
    if (ctx->d_pos) { e = DFREE(ctx->d_pos); e = DFREE(ctx->d_invexpo); e = DFREE(ctx->d_search); e = DFREE(ctx->d_errors); (void)e; }
    int cap = std::max(n, 512);
    HIPCHK(DMALLOC((void **)&ctx->d_pos, (size_t)cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_invexpo, (size_t)cap * 8));
    HIPCHK(DMALLOC((void **)&ctx->d_search, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_errors, (size_t)cap * 4));
    ctx->M_cap = cap;
  }
  if ((rc = ensure(ctx, ctx->d_warp, ctx->warp_cap, std::max((size_t)n * L * 64, (size_t)64)))) return rc;
  if ((rc = ensure(ctx, ctx->d_partials, ctx->partials_cap, std::max((size_t)visual_grid_inverse(std::max(n, 1)) * VIS_PSTRIDE, (size_t)64)))) return rc;
  return LIVO2_OK;
}

// tail stage: warp_map leaders (!normal_en with ids), k_warp_candidates, scan of the accept flags, gather into the frame arrays.  n = number of
// candidates or, with n_dev, its upper bound (the count itself is read on the device).  The candidate arrays d_c_* are filled already.
static int tail_enqueue(livo2_ctx *ctx, const livo2_retrieve_cfg *cfg, int width, int height, int stride, const uint8_t *d_ref_imgs, int n, const int32_t *n_dev,
                        bool have_ids, const int32_t *cand_point, const int32_t *cand_obs) {
  const int L = cfg->patch_pyrimid_level;
  const bool lead = have_ids && !cfg->normal_en;
  if (lead) {
    size_t ld = 1024; while (ld < 2 * (size_t)std::max(n, 1)) ld <<= 1;
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ctx->d_ld_keys, LEADER_EMPTY, ld, ctx->stream));
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)ctx->d_ld_vals, LEADER_EMPTY, ld, ctx->stream));
    hipLaunchKernelGGL(k_leader_insert, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_c_id, n_dev, n, ctx->d_ld_keys, ctx->d_ld_vals, (uint32_t)(ld - 1));
    hipLaunchKernelGGL(k_leader_lookup, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_c_id, n_dev, n, ctx->d_ld_keys, ctx->d_ld_vals, (uint32_t)(ld - 1),
                       ctx->d_c_leader);
  }
  WarpKernelArgs a{};
  a.img = ctx->d_img; a.ref_imgs = d_ref_imgs; a.width = width; a.height = height; a.stride = stride; a.n = n; a.L = L;
  a.normal_en = cfg->normal_en; a.ncc_en = cfg->ncc_en; a.fx = cfg->cam.fx; a.fy = cfg->cam.fy; a.cx = cfg->cam.cx; a.cy = cfg->cam.cy;
  std::memcpy(a.d, cfg->cam.d, 40); a.distortion = cfg->cam.distortion;
  a.inv_expo_cur = cfg->inv_expo_cur; a.ncc_thre = cfg->ncc_thre; a.outlier_threshold = cfg->outlier_threshold;
  std::memcpy(a.R_cur, cfg->R_cur, 72); std::memcpy(a.t_cur, cfg->t_cur, 24);
  a.pos = ctx->d_c_pos; a.normal = ctx->d_c_normal; a.ref_px = ctx->d_c_px; a.ref_f = ctx->d_c_f; a.ref_R = ctx->d_c_R; a.ref_t = ctx->d_c_t; a.ref_inv_expo = ctx->d_c_ie;
  a.ref_img_idx = ctx->d_c_idx; a.ref_level = ctx->d_c_lvl; a.patch_all = ctx->d_c_patch; a.accepted = ctx->d_c_acc; a.search_level = ctx->d_c_sl;
  a.error = ctx->d_c_err; a.ncc = ctx->d_c_ncc; a.A = ctx->d_c_A; a.n_dev = n_dev; a.leader = lead ? ctx->d_c_leader : nullptr;
  const int grid = (n + WARP_WAVES - 1) / WARP_WAVES;
  hipLaunchKernelGGL(k_warp_candidates, dim3(grid), dim3(WARP_WAVES * LIVO2_WAVE), 0, ctx->stream, a);
  hipLaunchKernelGGL(k_warp_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_c_acc, n, n_dev, ctx->d_c_slot, ctx->d_c_count);
  hipLaunchKernelGGL(k_warp_gather, dim3(grid), dim3(WARP_WAVES * LIVO2_WAVE), 0, ctx->stream, ctx->d_c_slot, n, n_dev, L, ctx->d_c_patch, ctx->d_c_pos, ctx->d_c_sl,
                     ctx->d_c_ie, ctx->d_warp, ctx->d_pos, ctx->d_search, ctx->d_invexpo, cand_point, cand_obs, cand_point ? ctx->d_sub_point : nullptr,
                     cand_obs ? ctx->d_sub_obs : nullptr);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

// per-candidate outputs of the tail stage -> host (n candidates)
static int tail_fetch(livo2_ctx *ctx, const livo2_retrieve_out *out, int n, int L) {
  if (!out || n <= 0) return LIVO2_OK;
  if (out->accepted) HIPCHK(devalloc::memcpy_async(out->accepted, ctx->d_c_acc, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (out->search_level) HIPCHK(devalloc::memcpy_async(out->search_level, ctx->d_c_sl, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (out->error) HIPCHK(devalloc::memcpy_async(out->error, ctx->d_c_err, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (out->ncc) HIPCHK(devalloc::memcpy_async(out->ncc, ctx->d_c_ncc, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (out->A_cur_ref) HIPCHK(devalloc::memcpy_async(out->A_cur_ref, ctx->d_c_A, (size_t)n * 32, hipMemcpyDeviceToHost, ctx->stream));
  if (out->patch_wrap) HIPCHK(devalloc::memcpy_async(out->patch_wrap, ctx->d_c_patch, (size_t)n * L * 256, hipMemcpyDeviceToHost, ctx->stream));
  return LIVO2_OK;
}

int livo2_visual_retrieve_warp(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const uint8_t *ref_imgs, int32_t n_ref,
                               const livo2_retrieve_candidates *cand, const livo2_retrieve_cfg *cfg, livo2_retrieve_out *out, int32_t *n_accepted) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!img || width <= 0 || height <= 0 || stride < width) return fail(ctx, LIVO2_ERR_INVALID, "bad image");
  if (!cand || !cfg || !n_accepted) return fail(ctx, LIVO2_ERR_INVALID, "cand / cfg / n_accepted is NULL");
  const int n = cand->n, L = cfg->patch_pyrimid_level;
  if (n < 0 || L < 1 || L > LIVO2_MAX_LEVELS) return fail(ctx, LIVO2_ERR_INVALID, "bad candidate count or patch_pyrimid_level");
  if (cfg->cam.width != width || cfg->cam.height != height) return fail(ctx, LIVO2_ERR_INVALID, "camera size differs from the image");
  if (n > 0 && (!ref_imgs || n_ref < 1 || !cand->pos || !cand->normal || !cand->ref_img_idx || !cand->ref_px || !cand->ref_f || !cand->ref_R || !cand->ref_t ||
                !cand->ref_level || !cand->ref_inv_expo)) return fail(ctx, LIVO2_ERR_INVALID, "bad candidate arrays");
  for (int i = 0; i < n; i++) {
    if (cand->ref_img_idx[i] < 0 || cand->ref_img_idx[i] >= n_ref) return fail(ctx, LIVO2_ERR_INVALID, "ref_img_idx out of range");
    if (cand->ref_level[i] < 0 || cand->ref_level[i] > 8) return fail(ctx, LIVO2_ERR_RANGE, "ref_level out of [0,8]");
    if (cand->ref_id && cand->ref_id[i] == LEADER_EMPTY) return fail(ctx, LIVO2_ERR_RANGE, "ref_id must differ from INT32_MAX");
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t img_bytes = (size_t)stride * height;
  int rc = ensure(ctx, ctx->d_img, ctx->img_cap, img_bytes); if (rc) return rc;
  if (n > 0) { rc = ensure(ctx, ctx->d_ref_imgs, ctx->ref_img_cap, img_bytes * n_ref); if (rc) return rc; }
  if ((rc = tail_reserve(ctx, n, L))) return rc;
  HIPCHK(devalloc::memcpy_async(ctx->d_img, img, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  int32_t count = 0;
  ctx->retrieve_kernel_us = 0.0;
  if (n > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_ref_imgs, ref_imgs, img_bytes * n_ref, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_pos, cand->pos, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_normal, cand->normal, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_px, cand->ref_px, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_f, cand->ref_f, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_R, cand->ref_R, (size_t)n * 72, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_t, cand->ref_t, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_ie, cand->ref_inv_expo, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_idx, cand->ref_img_idx, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_c_lvl, cand->ref_level, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    if (cand->ref_id) HIPCHK(devalloc::memcpy_async(ctx->d_c_id, cand->ref_id, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
    if ((rc = tail_enqueue(ctx, cfg, width, height, stride, ctx->d_ref_imgs, n, nullptr, cand->ref_id != nullptr, nullptr, nullptr))) return rc;
    HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
    HIPCHK(devalloc::memcpy_async(&count, ctx->d_c_count, 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = tail_fetch(ctx, out, n, L))) return rc;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
    ctx->retrieve_kernel_us = 1e3 * ms;
  } else {
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  *n_accepted = count;
  ctx->width = width; ctx->height = height; ctx->stride = stride; ctx->M = count; ctx->L = L;
  ctx->has_frame = true;
  ctx->has_ref = false;
  return LIVO2_OK;
}
double livo2_visual_retrieve_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->retrieve_kernel_us : 0.0; }

// ---- visual sub-map retrieval, the whole function: selection -> reference-patch choice -> tail as one chain ----------------------------
int livo2_visual_obs_upload(livo2_ctx *ctx, const livo2_visual_obs *o) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!o) return fail(ctx, LIVO2_ERR_INVALID, "obs is NULL");
  if (!ctx->has_vmap) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_visual_map_upload has not been called");
  const int n = ctx->n_vm, m = o->n_obs;
  if (m < 0 || o->n_ref < 0 || !o->point_offset || !o->normal || !o->normal_initialized || !o->ref_patch) return fail(ctx, LIVO2_ERR_INVALID, "bad observation table");
  if (m > 0 && (!o->id || !o->img_idx || !o->px || !o->f || !o->R || !o->t || !o->level || !o->inv_expo || !o->patch || !o->ref_imgs || o->n_ref < 1 || o->width < 1 ||
                o->height < 1 || o->stride < o->width)) return fail(ctx, LIVO2_ERR_INVALID, "bad observation arrays / reference images");
  if (o->point_offset[0] != 0 || o->point_offset[n] != m) return fail(ctx, LIVO2_ERR_INVALID, "point_offset must run from 0 to n_obs");
  for (int i = 0; i < n; i++) {
    if (o->point_offset[i + 1] < o->point_offset[i]) return fail(ctx, LIVO2_ERR_INVALID, "point_offset must not decrease");
    const int r = o->ref_patch[i];
    if (r != -1 && (r < o->point_offset[i] || r >= o->point_offset[i + 1])) return fail(ctx, LIVO2_ERR_INVALID, "ref_patch is not an observation of its point");
  }
  for (int k = 0; k < m; k++) {
    if (o->img_idx[k] < 0 || o->img_idx[k] >= o->n_ref) return fail(ctx, LIVO2_ERR_INVALID, "img_idx out of range");
    if (o->level[k] < 0 || o->level[k] > 8) return fail(ctx, LIVO2_ERR_RANGE, "level out of [0,8]");
    if (o->id[k] == LEADER_EMPTY) return fail(ctx, LIVO2_ERR_RANGE, "id must differ from INT32_MAX");
  }
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  const size_t m1 = (size_t)std::max(m, 1), n1 = (size_t)std::max(n, 1), img_bytes = (size_t)o->stride * o->height;
  int rc;
  if ((rc = ensure(ctx, ctx->d_ob_off, ctx->ob_off_cap, (size_t)n + 1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_id, ctx->ob_id_cap, m1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_img, ctx->ob_img_cap, m1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_lvl, ctx->ob_lvl_cap, m1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_px, ctx->ob_px_cap, m1 * 2))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_f, ctx->ob_f_cap, m1 * 3))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_R, ctx->ob_R_cap, m1 * 9))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_t, ctx->ob_t_cap, m1 * 3))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_ie, ctx->ob_ie_cap, m1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_patch, ctx->ob_patch_cap, m1 * 64))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_normal, ctx->vm_normal_cap, n1 * 3))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_ninit, ctx->vm_ninit_cap, n1))) return rc;
  if ((rc = ensure(ctx, ctx->d_vm_refpatch, ctx->vm_refpatch_cap, n1))) return rc;
  if ((rc = ensure(ctx, ctx->d_ob_imgs, ctx->ob_imgs_cap, std::max(img_bytes * (size_t)o->n_ref, (size_t)64)))) return rc;
  HIPCHK(devalloc::memcpy_async(ctx->d_ob_off, o->point_offset, ((size_t)n + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
  if (n > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_vm_normal, o->normal, (size_t)n * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_vm_ninit, o->normal_initialized, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_vm_refpatch, o->ref_patch, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
  }
  if (m > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_id, o->id, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_img, o->img_idx, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_lvl, o->level, (size_t)m * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_px, o->px, (size_t)m * 16, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_f, o->f, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_R, o->R, (size_t)m * 72, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_t, o->t, (size_t)m * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_ie, o->inv_expo, (size_t)m * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_patch, o->patch, (size_t)m * 256, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_ob_imgs, o->ref_imgs, img_bytes * (size_t)o->n_ref, hipMemcpyHostToDevice, ctx->stream));
  }
  {                                                     // obs_ lists: fixed stride, identity ranges of the CSR table
    int longest = 0;
    for (int i = 0; i < n; i++) longest = std::max(longest, o->point_offset[i + 1] - o->point_offset[i]);
    int stride = 32; while (stride < longest) stride <<= 1;
    ctx->ob_stride = stride;
    if ((rc = ensure(ctx, ctx->d_ob_list, ctx->ob_list_cap, n1 * (size_t)stride))) return rc;
    if ((rc = ensure(ctx, ctx->d_ob_cnt, ctx->ob_cnt_cap, n1))) return rc;
    if (n > 0) hipLaunchKernelGGL(k_ob_lists_from_csr, dim3((unsigned)(((size_t)n * stride + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ob_off, n, stride, ctx->d_ob_list, ctx->d_ob_cnt);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->n_obs = m; ctx->ob_n_ref = o->n_ref; ctx->ob_w = o->width; ctx->ob_h = o->height; ctx->ob_img_stride = o->stride;
  ctx->has_obs = true;
  return LIVO2_OK;
}

// ---- one frame's changes of the visual map: O(changes) ---------------------------------------------------------------------------------------------

int livo2_visual_map_counts(livo2_ctx *ctx, int32_t *c) {
  if (!ctx || !c) return LIVO2_ERR_INVALID;
  c[0] = ctx->has_vmap ? ctx->n_vm : 0; c[1] = ctx->has_obs ? ctx->n_obs : 0; c[2] = ctx->has_obs ? ctx->ob_n_ref : 0; c[3] = ctx->has_obs ? ctx->ob_stride : 0;
  return LIVO2_OK;
}

int livo2_visual_map_apply(livo2_ctx *ctx, const livo2_visual_map_delta *d) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!d) return fail(ctx, LIVO2_ERR_INVALID, "delta is NULL");
  if (!ctx->has_vmap || !ctx->has_obs) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_visual_map_upload + livo2_visual_obs_upload must have installed a map (an empty one will do)");
  const int np = d->n_new_points, no = d->n_new_obs, nt = d->n_touched;
  if (np < 0 || no < 0 || nt < 0) return fail(ctx, LIVO2_ERR_INVALID, "negative count");
  if ((np > 0 && (!d->new_pos || !d->new_voxel_key)) || (no > 0 && (!d->obs_id || !d->obs_img_idx || !d->obs_px || !d->obs_f || !d->obs_R || !d->obs_t || !d->obs_level || !d->obs_inv_expo || !d->obs_patch)) ||
      (nt > 0 && (!d->touched_point || !d->touched_offset || !d->touched_normal || !d->touched_normal_initialized || !d->touched_ref_patch)))
    return fail(ctx, LIVO2_ERR_INVALID, "a delta array is NULL");
  const int n0 = ctx->n_vm, m0 = ctx->n_obs, stride = ctx->ob_stride;
  const long long n1l = (long long)n0 + np, m1l = (long long)m0 + no;
  if (n1l > 0x7fffffffll / std::max(stride, 1) || m1l > 0x7fffffffll / 64) return fail(ctx, LIVO2_ERR_RANGE, "the visual map outgrows int32 indexing");
  const int n1 = (int)n1l, m1 = (int)m1l;
  int n_ref = ctx->ob_n_ref;
  if (d->img) {
    if (ctx->ob_w < 1 || ctx->ob_h < 1) return fail(ctx, LIVO2_ERR_INVALID, "no image geometry: the full upload must name width / height / stride (n_ref may be 0)");
    if (d->img_slot < 0 || d->img_slot > n_ref) return fail(ctx, LIVO2_ERR_INVALID, "img_slot must be an existing slot or the next free one");
    if (d->img_slot == n_ref) n_ref++;
  }
  const long long B = 1ll << 20;
  for (int i = 0; i < np; i++) {
    const int64_t *k = d->new_voxel_key + 3 * (size_t)i;
    if (k[0] < -B || k[0] >= B || k[1] < -B || k[1] >= B || k[2] < -B || k[2] >= B) return fail(ctx, LIVO2_ERR_RANGE, "visual voxel key outside 21 bits per axis");
  }
  for (int k = 0; k < no; k++) {
    if (d->obs_img_idx[k] < 0 || d->obs_img_idx[k] >= n_ref) return fail(ctx, LIVO2_ERR_INVALID, "img_idx out of range");
    if (d->obs_level[k] < 0 || d->obs_level[k] > 8) return fail(ctx, LIVO2_ERR_RANGE, "level out of [0,8]");
    if (d->obs_id[k] == LEADER_EMPTY) return fail(ctx, LIVO2_ERR_RANGE, "id must differ from INT32_MAX");
  }
  int n_tobs = 0;
  if (nt > 0) {
    if (d->touched_offset[0] != 0) return fail(ctx, LIVO2_ERR_INVALID, "touched_offset must start at 0");
    n_tobs = d->touched_offset[nt];
    if (n_tobs < 0 || (n_tobs > 0 && !d->touched_obs)) return fail(ctx, LIVO2_ERR_INVALID, "bad touched_obs");
    for (int q = 0; q < nt; q++) {
      const int p = d->touched_point[q], b = d->touched_offset[q], e = d->touched_offset[q + 1];
      if (p < 0 || p >= n1) return fail(ctx, LIVO2_ERR_INVALID, "touched_point out of range");
      if (e < b || e > n_tobs) return fail(ctx, LIVO2_ERR_INVALID, "touched_offset must not decrease");
      if (e - b > stride) return fail(ctx, LIVO2_ERR_RANGE, "an obs_ list is longer than the stride of the resident lists: re-upload the map (livo2_visual_obs_upload sizes the stride)");
      bool ref_ok = d->touched_ref_patch[q] == -1;
      for (int k = b; k < e; k++) {
        if (d->touched_obs[k] < 0 || d->touched_obs[k] >= m1) return fail(ctx, LIVO2_ERR_INVALID, "touched_obs out of range");
        ref_ok = ref_ok || d->touched_obs[k] == d->touched_ref_patch[q];
      }
      if (!ref_ok) return fail(ctx, LIVO2_ERR_INVALID, "ref_patch is not an observation of its point");
    }
  }
  HIPCHK(hipSetDevice(ctx->device));
  ctx->vm_delta_calls++;
  if (np == 0 && no == 0 && nt == 0 && !d->img) return LIVO2_OK;
  int rc;
  // ---- capacity (content kept)
  const size_t N0 = (size_t)n0, N1 = (size_t)std::max(n1, 1), M0 = (size_t)m0, M1 = (size_t)std::max(m1, 1), S = (size_t)stride;
  if ((rc = keep_grow(ctx, ctx->d_vm_pos, ctx->vm_pos_cap, N0 * 3, N1 * 3))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_pkey, ctx->vm_pkey_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_active, ctx->vm_active_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_fov, ctx->vm_fov_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_normal, ctx->vm_normal_cap, N0 * 3, N1 * 3))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_ninit, ctx->vm_ninit_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_vm_refpatch, ctx->vm_refpatch_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_list, ctx->ob_list_cap, N0 * S, N1 * S))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_cnt, ctx->ob_cnt_cap, N0, N1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_id, ctx->ob_id_cap, M0, M1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_img, ctx->ob_img_cap, M0, M1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_lvl, ctx->ob_lvl_cap, M0, M1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_px, ctx->ob_px_cap, M0 * 2, M1 * 2))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_f, ctx->ob_f_cap, M0 * 3, M1 * 3))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_R, ctx->ob_R_cap, M0 * 9, M1 * 9))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_t, ctx->ob_t_cap, M0 * 3, M1 * 3))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_ie, ctx->ob_ie_cap, M0, M1))) return rc;
  if ((rc = keep_grow(ctx, ctx->d_ob_patch, ctx->ob_patch_cap, M0 * 64, M1 * 64))) return rc;
  const size_t img_bytes = (size_t)ctx->ob_img_stride * ctx->ob_h;
  if (d->img && (rc = keep_grow(ctx, ctx->d_ob_imgs, ctx->ob_imgs_cap, img_bytes * (size_t)ctx->ob_n_ref, img_bytes * (size_t)n_ref))) return rc;
  // ---- one staging block
  Blob bl;
  const size_t o_pos = bl.add((size_t)np * 24), o_pkey = bl.add((size_t)np * 8), o_act = bl.add((size_t)np);
  const size_t o_oid = bl.add((size_t)no * 4), o_oimg = bl.add((size_t)no * 4), o_olvl = bl.add((size_t)no * 4), o_opx = bl.add((size_t)no * 16), o_of = bl.add((size_t)no * 24),
               o_oR = bl.add((size_t)no * 72), o_ot = bl.add((size_t)no * 24), o_oie = bl.add((size_t)no * 8), o_opatch = bl.add((size_t)no * 256);
  const size_t o_tp = bl.add((size_t)nt * 4), o_toff = bl.add(((size_t)nt + 1) * 4), o_tobs = bl.add((size_t)n_tobs * 4), o_tref = bl.add((size_t)nt * 4), o_tn = bl.add((size_t)nt * 24),
               o_tni = bl.add((size_t)nt), o_ta = bl.add((size_t)nt);
  const size_t o_img = bl.add(d->img ? img_bytes : 0);
  if (ctx->delta_ev_used) HIPCHK(hipEventSynchronize(ctx->delta_ev));          // the previous delta's copy has left the staging block
  if (bl.size > ctx->h_delta_cap) {
    if (ctx->h_delta) { hipError_t e = hipHostFree(ctx->h_delta); (void)e; ctx->h_delta = nullptr; }
    const size_t cap = std::max(bl.size, 2 * ctx->h_delta_cap + 4096);
    HIPCHK(hipHostMalloc(&ctx->h_delta, cap, hipHostMallocDefault));
    ctx->h_delta_cap = cap;
  }
  {
    size_t dcap = ctx->d_delta_cap; char *dp = (char *)ctx->d_delta;
    if ((rc = keep_grow(ctx, dp, dcap, 0, bl.size - (d->img ? ((img_bytes + 15) & ~(size_t)15) : 0) + 16))) return rc;
    ctx->d_delta = dp; ctx->d_delta_cap = dcap;
  }
  if (!ctx->delta_ev) HIPCHK(hipEventCreateWithFlags(&ctx->delta_ev, hipEventDisableTiming));
  char *h = (char *)ctx->h_delta;
  if (np) {
    std::memcpy(h + o_pos, d->new_pos, (size_t)np * 24);
    unsigned long long *pk = (unsigned long long *)(h + o_pkey);
    for (int i = 0; i < np; i++) { const int64_t *k = d->new_voxel_key + 3 * (size_t)i; pk[i] = ((unsigned long long)(k[0] + B) << 42) | ((unsigned long long)(k[1] + B) << 21) | (unsigned long long)(k[2] + B); }
    if (d->new_active) std::memcpy(h + o_act, d->new_active, (size_t)np); else std::memset(h + o_act, 1, (size_t)np);
  }
  if (no) {
    std::memcpy(h + o_oid, d->obs_id, (size_t)no * 4); std::memcpy(h + o_oimg, d->obs_img_idx, (size_t)no * 4); std::memcpy(h + o_olvl, d->obs_level, (size_t)no * 4);
    std::memcpy(h + o_opx, d->obs_px, (size_t)no * 16); std::memcpy(h + o_of, d->obs_f, (size_t)no * 24); std::memcpy(h + o_oR, d->obs_R, (size_t)no * 72);
    std::memcpy(h + o_ot, d->obs_t, (size_t)no * 24); std::memcpy(h + o_oie, d->obs_inv_expo, (size_t)no * 8); std::memcpy(h + o_opatch, d->obs_patch, (size_t)no * 256);
  }
  if (nt) {
    std::memcpy(h + o_tp, d->touched_point, (size_t)nt * 4); std::memcpy(h + o_toff, d->touched_offset, ((size_t)nt + 1) * 4);
    if (n_tobs) std::memcpy(h + o_tobs, d->touched_obs, (size_t)n_tobs * 4);
    std::memcpy(h + o_tref, d->touched_ref_patch, (size_t)nt * 4); std::memcpy(h + o_tn, d->touched_normal, (size_t)nt * 24);
    std::memcpy(h + o_tni, d->touched_normal_initialized, (size_t)nt);
    if (d->touched_active) std::memcpy(h + o_ta, d->touched_active, (size_t)nt); else std::memset(h + o_ta, 1, (size_t)nt);
  }
  if (d->img) std::memcpy(h + o_img, d->img, img_bytes);
  if (o_img > 0) HIPCHK(devalloc::memcpy_async(ctx->d_delta, h, o_img, hipMemcpyHostToDevice, ctx->stream));
  if (d->img) HIPCHK(devalloc::memcpy_async(ctx->d_ob_imgs + img_bytes * (size_t)d->img_slot, h + o_img, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->delta_ev, ctx->stream)); ctx->delta_ev_used = true;
  const char *g = (const char *)ctx->d_delta;
  VmDeltaArgs a{};
  a.n_points = n0; a.n_obs = m0; a.n_new_points = np; a.n_new_obs = no; a.n_touched = nt; a.stride = stride;
  a.s_pos = (const double *)(g + o_pos); a.s_pkey = (const unsigned long long *)(g + o_pkey); a.s_active = (const uint8_t *)(g + o_act);
  a.s_oid = (const int32_t *)(g + o_oid); a.s_oimg = (const int32_t *)(g + o_oimg); a.s_olvl = (const int32_t *)(g + o_olvl); a.s_opx = (const double *)(g + o_opx);
  a.s_of = (const double *)(g + o_of); a.s_oR = (const double *)(g + o_oR); a.s_ot = (const double *)(g + o_ot); a.s_oie = (const double *)(g + o_oie); a.s_opatch = (const float *)(g + o_opatch);
  a.s_tpoint = (const int32_t *)(g + o_tp); a.s_toff = (const int32_t *)(g + o_toff); a.s_tobs = (const int32_t *)(g + o_tobs); a.s_trefp = (const int32_t *)(g + o_tref);
  a.s_tnormal = (const double *)(g + o_tn); a.s_tninit = (const uint8_t *)(g + o_tni); a.s_tactive = (const uint8_t *)(g + o_ta);
  a.pos = ctx->d_vm_pos; a.pkey = ctx->d_vm_pkey; a.active = ctx->d_vm_active; a.fov = ctx->d_vm_fov;
  a.oid = ctx->d_ob_id; a.oimg = ctx->d_ob_img; a.olvl = ctx->d_ob_lvl; a.opx = ctx->d_ob_px; a.of = ctx->d_ob_f; a.oR = ctx->d_ob_R; a.ot = ctx->d_ob_t; a.oie = ctx->d_ob_ie; a.opatch = ctx->d_ob_patch;
  a.list = ctx->d_ob_list; a.count = ctx->d_ob_cnt; a.refp = ctx->d_vm_refpatch; a.normal = ctx->d_vm_normal; a.ninit = ctx->d_vm_ninit;
  const size_t work = std::max((size_t)np, (size_t)no * 32);
  if (work) hipLaunchKernelGGL(k_vm_apply, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx->stream, a);
  if (nt) hipLaunchKernelGGL(k_vm_apply_touched, dim3((unsigned)(((size_t)nt * stride + 255) / 256)), dim3(256), 0, ctx->stream, a);
  HIPCHK(hipGetLastError());
  // the voxel set of the RayCasting module follows the points (rebuilt larger when it would pass half full)
  if (np > 0 && ctx->d_vm_set) {
    if (2 * (size_t)n1 > ctx->vm_set_cap) {
      size_t cap = ctx->vm_set_cap; while (cap < 4 * (size_t)n1) cap <<= 1;
      if ((rc = ensure(ctx, ctx->d_vm_set, ctx->vm_set_cap, cap))) return rc;
      ctx->vm_set_cap = cap; ctx->vm_set_mask = (uint32_t)(cap - 1);
      HIPCHK(hipMemsetAsync(ctx->d_vm_set, 0xFF, cap * 8, ctx->stream));
      hipLaunchKernelGGL(k_vm_voxel_set, dim3((n1 + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_vm_pkey, n1, ctx->d_vm_set, ctx->vm_set_mask);
    } else hipLaunchKernelGGL(k_vm_voxel_set, dim3((np + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_vm_pkey + n0, np, ctx->d_vm_set, ctx->vm_set_mask);
    HIPCHK(hipGetLastError());
  }
  ctx->n_vm = n1; ctx->n_obs = m1; ctx->ob_n_ref = n_ref;
  return LIVO2_OK;
}

int livo2_visual_retrieve_from_map(livo2_ctx *ctx, const uint8_t *img, int32_t width, int32_t height, int32_t stride, const double *pg, int32_t n_pg,
                                   const livo2_select_cfg *sel, const livo2_retrieve_cfg *cfg, livo2_retrieve_chain_out *out, int32_t *n_candidates, int32_t *n_accepted) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!img || width <= 0 || height <= 0 || stride < width) return fail(ctx, LIVO2_ERR_INVALID, "bad image");
  const bool pg_resident = !pg && n_pg == LIVO2_PG_FROM_MAP_UPDATE;      // the pv_list_ of the last livo2_map_tree_update[_from_scan] (LIVMapper.cpp:413-426: _pv_list)
  if (pg_resident) {
    if (ctx->mt_pv_n < 0) return fail(ctx, LIVO2_ERR_NO_SCAN, "LIVO2_PG_FROM_MAP_UPDATE: no livo2_map_tree_update[_from_scan] since the last livo2_lidar_set_scan");
    n_pg = ctx->mt_pv_n;
  }
  if (!sel || !cfg || !n_accepted || n_pg < 0 || (n_pg > 0 && !pg && !pg_resident)) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (!ctx->has_obs) return fail(ctx, LIVO2_ERR_NO_MAP, "livo2_visual_obs_upload has not been called (after livo2_visual_map_upload)");
  const int L = cfg->patch_pyrimid_level;
  if (L < 1 || L > LIVO2_MAX_LEVELS) return fail(ctx, LIVO2_ERR_INVALID, "bad patch_pyrimid_level");
  if (cfg->cam.width != width || cfg->cam.height != height || sel->cam.width != width || sel->cam.height != height) return fail(ctx, LIVO2_ERR_INVALID, "camera size differs from the image");
  if (ctx->n_obs > 0 && (ctx->ob_w != width || ctx->ob_h != height || ctx->ob_img_stride != stride)) return fail(ctx, LIVO2_ERR_INVALID, "reference images differ in size from the image");
  if (sel->border < 4) return fail(ctx, LIVO2_ERR_INVALID, "border must keep the 8x8 patch of a selected point inside the image (>= 4)");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  size_t cap = 0;
  int rc = select_reserve(ctx, sel, n_pg, &cap); if (rc) return rc;
  const int length = sel->grid_n_width * sel->grid_n_height;
  const size_t len = (size_t)length, img_bytes = (size_t)stride * height;
  if ((rc = ensure(ctx, ctx->d_img, ctx->img_cap, img_bytes))) return rc;
  if ((rc = tail_reserve(ctx, length, L))) return rc;
  if ((rc = ensure(ctx, ctx->d_ch_obs, ctx->ch_obs_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_ch_flag, ctx->ch_flag_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_ch_slot, ctx->ch_slot_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_cand_cell, ctx->cand_cell_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_cand_point, ctx->cand_point_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_cand_obs, ctx->cand_obs_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_sub_point, ctx->sub_point_cap, len))) return rc;
  if ((rc = ensure(ctx, ctx->d_sub_obs, ctx->sub_obs_cap, len))) return rc;
  if (!ctx->d_ch_count) HIPCHK(DMALLOC((void **)&ctx->d_ch_count, 64));
  HIPCHK(devalloc::memcpy_async(ctx->d_img, img, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (n_pg > 0) HIPCHK(devalloc::memcpy_async(ctx->d_sel_pg, pg_resident ? ctx->mt_in_pw : pg, (size_t)n_pg * 24, pg_resident ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipEventRecord(ctx->span0, ctx->stream));
  // 1. selection
  if ((rc = select_enqueue(ctx, sel, n_pg, cap))) return rc;
  // 2. reference-patch choice, then the chosen pairs lined up as candidates in grid-cell order
  ChoiceArgs ca{};
  ca.normal_en = cfg->normal_en; ca.length = length;
  frame_pos(sel->R_cur, sel->t_cur, ca.cam_pos);
  ca.cell_point = ctx->d_sel_point; ca.cell_discont = ctx->d_sel_disc; ca.pos = ctx->d_vm_pos; ca.obs_list = ctx->d_ob_list; ca.obs_count = ctx->d_ob_cnt; ca.obs_stride = ctx->ob_stride; ca.obs_id = ctx->d_ob_id;
  ca.obs_R = ctx->d_ob_R; ca.obs_t = ctx->d_ob_t; ca.obs_patch = ctx->d_ob_patch; ca.normal_init = ctx->d_vm_ninit; ca.ref_patch = ctx->d_vm_refpatch;
  ca.cell_obs = ctx->d_ch_obs; ca.cell_flag = ctx->d_ch_flag;
  hipLaunchKernelGGL(k_choose_ref, dim3((length + CHOICE_WAVES - 1) / CHOICE_WAVES), dim3(CHOICE_WAVES * LIVO2_WAVE), 0, ctx->stream, ca);
  hipLaunchKernelGGL(k_warp_scan, dim3(1), dim3(1024), 0, ctx->stream, ctx->d_ch_flag, length, (const int32_t *)nullptr, ctx->d_ch_slot, ctx->d_ch_count);
  GatherCandArgs ga{};
  ga.length = length; ga.slot = ctx->d_ch_slot; ga.cell_point = ctx->d_sel_point; ga.cell_obs = ctx->d_ch_obs; ga.pos = ctx->d_vm_pos; ga.normal = ctx->d_vm_normal;
  ga.obs_id = ctx->d_ob_id; ga.obs_img_idx = ctx->d_ob_img; ga.obs_level = ctx->d_ob_lvl; ga.obs_px = ctx->d_ob_px; ga.obs_f = ctx->d_ob_f; ga.obs_R = ctx->d_ob_R;
  ga.obs_t = ctx->d_ob_t; ga.obs_inv_expo = ctx->d_ob_ie;
  ga.c_pos = ctx->d_c_pos; ga.c_normal = ctx->d_c_normal; ga.c_px = ctx->d_c_px; ga.c_f = ctx->d_c_f; ga.c_R = ctx->d_c_R; ga.c_t = ctx->d_c_t; ga.c_ie = ctx->d_c_ie;
  ga.c_idx = ctx->d_c_idx; ga.c_lvl = ctx->d_c_lvl; ga.c_id = ctx->d_c_id; ga.cand_cell = ctx->d_cand_cell; ga.cand_point = ctx->d_cand_point; ga.cand_obs = ctx->d_cand_obs;
  hipLaunchKernelGGL(k_gather_candidates, dim3((length + 7) / 8), dim3(256), 0, ctx->stream, ga);
  HIPCHK(hipGetLastError());
  // 3. tail over the candidates (their number stays on the device), survivors -> the resident frame
  if ((rc = tail_enqueue(ctx, cfg, width, height, stride, ctx->d_ob_imgs, length, ctx->d_ch_count, true, ctx->d_cand_point, ctx->d_cand_obs))) return rc;
  HIPCHK(hipEventRecord(ctx->span1, ctx->stream));
  int32_t counts[2] = {0, 0}, flag = 0;
  HIPCHK(devalloc::memcpy_async(&counts[0], ctx->d_ch_count, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(&counts[1], ctx->d_c_count, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(&flag, ctx->d_sel_flag, 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, ctx->span0, ctx->span1));
  ctx->chain_kernel_us = 1e3 * ms;
  if (flag) return fail(ctx, LIVO2_ERR_RANGE, "scan voxel key outside 21 bits per axis");
  const int nc = counts[0], na = counts[1];
  if (out) {
    if (out->cell_point) HIPCHK(devalloc::memcpy_async(out->cell_point, ctx->d_sel_point, len * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->cell_dist) HIPCHK(devalloc::memcpy_async(out->cell_dist, ctx->d_sel_dist, len * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->cell_discontinuous) HIPCHK(devalloc::memcpy_async(out->cell_discontinuous, ctx->d_sel_disc, len, hipMemcpyDeviceToHost, ctx->stream));
    if (out->cell_obs) HIPCHK(devalloc::memcpy_async(out->cell_obs, ctx->d_ch_obs, len * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->ref_patch && ctx->n_vm > 0) HIPCHK(devalloc::memcpy_async(out->ref_patch, ctx->d_vm_refpatch, (size_t)ctx->n_vm * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->cand_cell && nc > 0) HIPCHK(devalloc::memcpy_async(out->cand_cell, ctx->d_cand_cell, (size_t)nc * 4, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = tail_fetch(ctx, &out->tail, nc, L))) return rc;
    if (out->sub_point && na > 0) HIPCHK(devalloc::memcpy_async(out->sub_point, ctx->d_sub_point, (size_t)na * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out->sub_obs && na > 0) HIPCHK(devalloc::memcpy_async(out->sub_obs, ctx->d_sub_obs, (size_t)na * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  if (n_candidates) *n_candidates = nc;
  *n_accepted = na;
  ctx->width = width; ctx->height = height; ctx->stride = stride; ctx->M = na; ctx->L = L;
  ctx->has_frame = true;
  ctx->has_ref = false;
  return LIVO2_OK;
}
int livo2_visual_raycast_fetch(livo2_ctx *ctx, double *center_normal, int32_t capacity, int32_t *n) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!n || capacity < 0 || (capacity > 0 && !center_normal)) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  if (ctx->ray_add_n < 0) return fail(ctx, LIVO2_ERR_INVALID, "the last selection ran without raycast_en");
  HIPCHK(hipSetDevice(ctx->device));
  int32_t cnt[2] = {0, 0};
  HIPCHK(devalloc::memcpy_async(cnt, ctx->d_ray_counters, 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  *n = cnt[1];
  const int m = std::min(cnt[1], capacity);
  if (m > 0) { HIPCHK(devalloc::memcpy_async(center_normal, ctx->d_ray_add, (size_t)m * 48, hipMemcpyDeviceToHost, ctx->stream)); HIPCHK(hipStreamSynchronize(ctx->stream)); }
  return LIVO2_OK;
}

double livo2_visual_retrieve_from_map_last_kernel_us(const livo2_ctx *ctx) { return ctx ? ctx->chain_kernel_us : 0.0; }

static int visual_ready(livo2_ctx *ctx, const livo2_state *a, const livo2_state *b, const livo2_visual_cfg *cfg) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!a || !b) return fail(ctx, LIVO2_ERR_INVALID, "state is NULL");
  if (!ctx->has_frame) return fail(ctx, LIVO2_ERR_NO_FRAME, "livo2_visual_set_frame has not been called");
  int rc = check_visual_cfg(ctx, cfg); if (rc) return rc;
  HIPCHK(hipSetDevice(ctx->device));
  return LIVO2_OK;
}

int livo2_visual_iterate(livo2_ctx *ctx, int32_t level, const livo2_state *cur, const livo2_visual_cfg *cfg, livo2_visual_sums *sums, float *errors, double *z,
                         double *H_sub) {
  int rc = visual_ready(ctx, cur, cur, cfg); if (rc) return rc;
  if (!sums) return fail(ctx, LIVO2_ERR_INVALID, "sums is NULL");
  if (level < 0 || level >= cfg->patch_pyrimid_level) return fail(ctx, LIVO2_ERR_INVALID, "level out of range");
  const int M = ctx->M;
  if ((z || H_sub) && ctx->dbg_cap < M) {
    hipError_t e; if (ctx->d_zdbg) { e = DFREE(ctx->d_zdbg); e = DFREE(ctx->d_Hdbg); (void)e; }
    HIPCHK(DMALLOC((void **)&ctx->d_zdbg, (size_t)std::max(M, 1) * 64 * 8)); HIPCHK(DMALLOC((void **)&ctx->d_Hdbg, (size_t)std::max(M, 1) * 64 * 56));
    ctx->dbg_cap = M;
  }
  rc = upload_states(ctx, cur, cur); if (rc) return rc;
  VisualKernelArgs a = make_visual_args(ctx, cfg, level);
  a.errors = ctx->d_errors; a.z = z ? ctx->d_zdbg : nullptr; a.H_sub = H_sub ? ctx->d_Hdbg : nullptr;
  const int grid = cfg->inverse_composition_en ? visual_grid_inverse(std::max(M, 1)) : visual_grid(std::max(M, 1));
  if (z) HIPCHK(hipMemsetAsync(ctx->d_zdbg, 0, (size_t)std::max(M, 1) * 64 * 8, ctx->stream));       // a skipped (out-of-image) patch leaves its rows untouched
  if (H_sub) HIPCHK(hipMemsetAsync(ctx->d_Hdbg, 0, (size_t)std::max(M, 1) * 64 * 56, ctx->stream));
  if (cfg->inverse_composition_en && M > 0) {
    VisualRefArgs r = make_ref_args(ctx);
    hipLaunchKernelGGL(k_visual_ref_precompute, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, r);
    Timed t(ctx, 1);
    if (z || H_sub) hipLaunchKernelGGL(k_visual_inverse_residual<true>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, r, ctx->d_ctl, ctx->d_partials, 0);
    else hipLaunchKernelGGL(k_visual_inverse_residual<false>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, r, ctx->d_ctl, ctx->d_partials, 0);
    t.done();
  } else if (z || H_sub) {
    Timed t(ctx, 1); hipLaunchKernelGGL(k_visual_residual<true>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, ctx->d_ctl, ctx->d_partials, 0); t.done();
  } else {
    Timed t(ctx, 1); hipLaunchKernelGGL(k_visual_residual<false>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, ctx->d_ctl, ctx->d_partials, 0); t.done();
  }
  { Timed t(ctx, 3); hipLaunchKernelGGL(k_visual_solve, dim3(1), dim3(512), 0, ctx->stream, ctx->d_ctl, ctx->d_partials, grid, 0, level, 0, cfg->img_point_cov, visual_solve_args(ctx, cfg)); t.done(); }
  HIPCHK(hipGetLastError());
  HIPCHK(devalloc::memcpy_async(ctx->h_out, &ctx->d_ctl->sums_v, sizeof(livo2_visual_sums), hipMemcpyDeviceToHost, ctx->stream));
  if (errors && M > 0) HIPCHK(devalloc::memcpy_async(errors, ctx->d_errors, (size_t)M * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (z && M > 0) HIPCHK(devalloc::memcpy_async(z, ctx->d_zdbg, (size_t)M * 64 * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (H_sub && M > 0) HIPCHK(devalloc::memcpy_async(H_sub, ctx->d_Hdbg, (size_t)M * 64 * 56, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memcpy(sums, ctx->h_out, sizeof(livo2_visual_sums));
  return LIVO2_OK;
}

// the iterate and the prior of a visual update := the LiDAR posterior that the update before it left in the result block (LIVMapper.cpp:135-136, 256, 371: `state`
// is shared and processImu re-assigns state_propagat from it), header cleared as upload_states clears it — the hand-over of livo2_frame_update, all on the device
__global__ void __launch_bounds__(256) k_ctl_chain_visual(DevCtl *__restrict__ ctl) {
  const int t = threadIdx.x;
  const double *src = reinterpret_cast<const double *>(&ctl->lidar.state);
  double *c = reinterpret_cast<double *>(&ctl->cur), *p = reinterpret_cast<double *>(&ctl->prop);
  for (int k = t; k < (int)(sizeof(livo2_state) / 8); k += 256) { const double v = src[k]; c[k] = v; p[k] = v; }
  if (t == 0) { ctl->hdr.stop = 0; ctl->hdr.rematch_num = 0; ctl->hdr.reserved = 0; ctl->hdr.last_error = FLT_MAX; ctl->hdr.n_steps = 0; ctl->hdr.pad[0] = ctl->hdr.pad[1] = ctl->hdr.pad[2] = 0; }
  if (t < 9) ctl->hdr.RE[t] = 0.0;
}

static int visual_enqueue(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg, int level_hi, int level_lo,
                          int iters, int mode) {
  int rc;
  if (state_in) { rc = upload_states(ctx, state_in, prop); if (rc) return rc; }
  else { hipLaunchKernelGGL(k_ctl_chain_visual, dim3(1), dim3(256), 0, ctx->stream, ctx->d_ctl); HIPCHK(hipGetLastError()); }      // chained: states from the LiDAR result block
  const int grid = cfg->inverse_composition_en ? visual_grid_inverse(std::max(ctx->M, 1)) : visual_grid(std::max(ctx->M, 1));
  VisualKernelArgs a{};
  ctx->vp_last_valid = false;
  bool want_persistent = mode == 1 && !cfg->inverse_composition_en && ctx->visual_persistent && !ctx->vp_rerun && level_lo == 0 && level_hi == cfg->patch_pyrimid_level - 1 && iters == cfg->max_iterations;
  if (want_persistent && ctx->vp_backoff_left > 0) { ctx->vp_backoff_left--; ctx->vp_backoff_skips++; want_persistent = false; }     // a recent grid timed out: stay on the per-step path for a while
  if (want_persistent) {
    {                        // (before the admission: nothing between the reservation and the launch may fail) exchange buffers start as all-zero words: tag 0 is never a step's tag
      const size_t c0 = ctx->vp_rows_cap, c1 = ctx->vp_errs_cap;
      rc = ensure(ctx, ctx->d_vp_rows, ctx->vp_rows_cap, (size_t)2 * VP_MAX_ROWS * VIS_PSTRIDE * 2); if (rc) return rc;
      rc = ensure(ctx, ctx->d_vp_errs, ctx->vp_errs_cap, (size_t)2 * std::max(ctx->M_cap, 512)); if (rc) return rc;
      if (ctx->vp_rows_cap != c0) HIPCHK(hipMemsetAsync(ctx->d_vp_rows, 0, ctx->vp_rows_cap * 8, ctx->stream));
      if (ctx->vp_errs_cap != c1) HIPCHK(hipMemsetAsync(ctx->d_vp_errs, 0, ctx->vp_errs_cap * 8, ctx->stream));
      if (ctx->vp_prof && !ctx->d_vp_prof) HIPCHK(DMALLOC((void **)&ctx->d_vp_prof, VP_MAX_BLOCKS * 32 * 16 * 8));
      if (ctx->vp_prof) HIPCHK(hipMemsetAsync(ctx->d_vp_prof, 0, VP_MAX_BLOCKS * 32 * 16 * 8, ctx->stream));
    }
    int halves = 1;
    const int G = persist_admit(ctx, grid, &halves);
    if (G > 0) {
      VisPersistArgs p{};
      p.a = make_visual_args(ctx, cfg, 0);
      p.a.errors = ctx->d_errors;
      p.rows = ctx->d_vp_rows; p.errs = ctx->d_vp_errs;
      p.n_rows = std::min(grid, VP_MAX_ROWS); p.halves = halves;
      p.levels = cfg->patch_pyrimid_level; p.max_iterations = cfg->max_iterations; p.error_threads = cfg->mp_proc_num; p.img_point_cov = cfg->img_point_cov;
      ctx->vp_seq = (ctx->vp_seq + 1) & 0xffffffu; if (ctx->vp_seq == 0) ctx->vp_seq = 1;       // tag 0 = never-written memory
      p.tag_base = ctx->vp_seq << 8;
      if (ctx->vp_prof) p.prof = ctx->d_vp_prof;
      p.timeout = ctx->vp_debug_timeout ? 200000ull : (unsigned long long)ctx->vp_timeout_us * 100ull;      // 100 MHz ticks; debug: 2 ms
      p.debug_drop_block = (ctx->vp_debug_timeout && G > 1) ? G - 1 : -1;
      ctx->vp_last_chained = state_in == nullptr;                                                                      // what a timed-out grid is re-run from
      if (state_in) { ctx->vp_last_in = *state_in; ctx->vp_last_prop = *prop; }
      ctx->vp_last_cfg = *cfg; ctx->vp_last_valid = true;
      { Timed t(ctx, 1); hipLaunchKernelGGL(k_visual_update_persistent, dim3(G), dim3(VP_BLOCK), 0, ctx->stream, p, ctx->d_ctl); t.done(); }
      const hipError_t le = hipGetLastError();
      persist_register(ctx, le == hipSuccess ? G : 0);            // a failed launch gives its reservation back
      HIPCHK(le);
      ctx->vp_used++;
      return LIVO2_OK;
    }
    ctx->vp_fallback++;
  }
  for (int level = level_hi; level >= level_lo; level--) {
    a = make_visual_args(ctx, cfg, level);
    a.errors = ctx->d_errors;
    const bool inverse = cfg->inverse_composition_en != 0;
    VisualRefArgs r{};
    if (inverse) {             // has_ref_patch_cache = false at every level (vio.cpp:794-795)
      r = make_ref_args(ctx);
      hipLaunchKernelGGL(k_visual_ref_precompute, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, r);
    }
    for (int it = 0; it < iters; it++) {
      {
        Timed t(ctx, 1);
        if (inverse) hipLaunchKernelGGL(k_visual_inverse_residual<false>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, r, ctx->d_ctl, ctx->d_partials, (mode == 1 && it > 0) ? 1 : 0);
        else hipLaunchKernelGGL(k_visual_residual<false>, dim3(grid), dim3(VIS_BLOCK), 0, ctx->stream, a, ctx->d_ctl, ctx->d_partials, (mode == 1 && it > 0) ? 1 : 0);
        t.done();
      }
      { Timed t(ctx, 3); hipLaunchKernelGGL(k_visual_solve, dim3(1), dim3(512), 0, ctx->stream, ctx->d_ctl, ctx->d_partials, grid, mode, level, mode == 1 ? it : (it == 0 ? 0 : 1), cfg->img_point_cov, visual_solve_args(ctx, cfg)); t.done(); }
    }
  }
  hipLaunchKernelGGL(k_visual_finish, dim3(1), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->d_ctl, a, mode == 1 ? 1 : 0);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

int livo2_visual_update_async(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg) {
  int rc = visual_ready(ctx, state_in, prop, cfg); if (rc) return rc;
  if (ctx->M == 0) {            // total_points == 0: computeJacobianAndUpdateEKF returns immediately (vio.cpp:786)
    rc = upload_states(ctx, state_in, prop); if (rc) return rc;
    VisualKernelArgs a = make_visual_args(ctx, cfg, 0);
    hipLaunchKernelGGL(k_visual_finish, dim3(1), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->d_ctl, a, 0);
    HIPCHK(hipGetLastError());
    return LIVO2_OK;
  }
  return visual_enqueue(ctx, state_in, prop, cfg, cfg->patch_pyrimid_level - 1, 0, cfg->max_iterations, 1);
}

int livo2_visual_update_fetch(livo2_ctx *ctx, livo2_visual_result *result, float *errors) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!result) return fail(ctx, LIVO2_ERR_INVALID, "result is NULL");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(devalloc::memcpy_async(ctx->h_out, &ctx->d_ctl->visual, sizeof(livo2_visual_result), hipMemcpyDeviceToHost, ctx->stream));
  if (errors && ctx->M > 0) HIPCHK(devalloc::memcpy_async(errors, ctx->d_errors, (size_t)ctx->M * 4, hipMemcpyDeviceToHost, ctx->stream));
  int32_t *flag = reinterpret_cast<int32_t *>(static_cast<char *>(ctx->h_out) + sizeof(livo2_visual_result));
  HIPCHK(devalloc::memcpy_async(flag, &ctx->d_ctl->hdr.pad[0], 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (*flag) {
    // the resident grid gave up (a block was not co-resident: another process or a long foreign kernel held its CU) and committed nothing.  Re-run the update as the
    // launch-per-step sequence from the inputs kept at enqueue; the caller sees the same result, "visual_persistent_timeouts" counts the event.
    if (!ctx->vp_last_valid) return fail(ctx, LIVO2_ERR_HIP, "persistent visual update timed out and its inputs are gone");
    ctx->vp_timeouts++;
    ctx->vp_backoff_len = ctx->vp_backoff_len ? std::min(2 * ctx->vp_backoff_len, 1024) : 8;       // sustained contention: back off longer each time
    ctx->vp_backoff_left = ctx->vp_debug_timeout ? 0 : ctx->vp_backoff_len;                         // (the test hook times out on purpose, every time)
    HIPCHK(hipMemsetAsync(&ctx->d_ctl->hdr.pad[0], 0, 4, ctx->stream));
    const livo2_state in = ctx->vp_last_in, pr = ctx->vp_last_prop; const livo2_visual_cfg vc = ctx->vp_last_cfg;
    ctx->vp_rerun = true;
    const int rc = visual_enqueue(ctx, ctx->vp_last_chained ? nullptr : &in, ctx->vp_last_chained ? nullptr : &pr, &vc, vc.patch_pyrimid_level - 1, 0, vc.max_iterations, 1);
    ctx->vp_rerun = false;
    if (rc) return rc;
    HIPCHK(devalloc::memcpy_async(ctx->h_out, &ctx->d_ctl->visual, sizeof(livo2_visual_result), hipMemcpyDeviceToHost, ctx->stream));
    if (errors && ctx->M > 0) HIPCHK(devalloc::memcpy_async(errors, ctx->d_errors, (size_t)ctx->M * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  }
  else if (ctx->vp_backoff_left == 0) ctx->vp_backoff_len = 0;     // a resident grid completed: the next time-out starts the back-off from 8 again
  std::memcpy(result, ctx->h_out, sizeof(livo2_visual_result));
  return rz_gate(ctx);
}

int livo2_visual_update(livo2_ctx *ctx, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg, livo2_visual_result *result,
                        float *errors) {
  int rc = livo2_visual_update_async(ctx, state_in, prop, cfg); if (rc) return rc;
  return livo2_visual_update_fetch(ctx, result, errors);
}

// ---- one LIO + VIO frame per call -------------------------------------------------------------------------------------------------------------------
namespace {
struct FrameRes { livo2_lidar_result lidar; livo2_visual_result visual; int32_t timed_out, pad; };
}
int livo2_frame_update_async(livo2_ctx *ctx, const livo2_frame_in *f) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!f || !f->prior || !f->lidar_cfg || !f->visual_cfg) return fail(ctx, LIVO2_ERR_INVALID, "frame / prior / cfg is NULL");
  if (ctx->frame_inflight >= 2) return fail(ctx, LIVO2_ERR_INVALID, "two frames are in flight: livo2_frame_update_fetch first");
  int rc = check_visual_cfg(ctx, f->visual_cfg); if (rc) return rc;
  const int M = f->M, L = f->L;
  if (!f->img || f->width <= 0 || f->height <= 0 || f->stride < f->width) return fail(ctx, LIVO2_ERR_INVALID, "bad image");
  if (M < 0 || L < 1 || L > LIVO2_MAX_LEVELS || (M > 0 && (!f->pos || !f->warp_patch || !f->search_levels || !f->inv_expo_list))) return fail(ctx, LIVO2_ERR_INVALID, "bad sub-map arrays");
  if (L != f->visual_cfg->patch_pyrimid_level) return fail(ctx, LIVO2_ERR_INVALID, "warp_patch holds L levels per patch: L must equal visual_cfg->patch_pyrimid_level");
  for (int i = 0; i < M; i++) if (f->search_levels[i] < 0 || f->search_levels[i] > 8) return fail(ctx, LIVO2_ERR_RANGE, "search_level out of [0,8]");
  // ---- LIO: scan up, StateEstimation from the prior (both asynchronous)
  rc = livo2_lidar_set_scan(ctx, f->xyz, f->n_points, f->lidar_cfg); if (rc) return rc;
  rc = lidar_ready(ctx, f->prior, f->prior, f->lidar_cfg); if (rc) return rc;
  rc = ensure_lidar_outputs(ctx, nullptr); if (rc) return rc;
  rc = lidar_enqueue(ctx, f->prior, f->prior, f->lidar_cfg, f->lidar_cfg->max_iterations, 1); if (rc) return rc;
  // ---- the frame: image + sub-map through a pinned staging block (the caller's arrays are free on return; device buffers grow only between frames in flight)
  const size_t img_bytes = (size_t)f->stride * f->height;
  const size_t o_pos = (img_bytes + 15) & ~(size_t)15, o_warp = o_pos + (((size_t)M * 24 + 15) & ~(size_t)15), o_sl = o_warp + (((size_t)M * L * 256 + 15) & ~(size_t)15),
               o_ie = o_sl + (((size_t)M * 4 + 15) & ~(size_t)15), total = o_ie + (size_t)M * 8;
  const bool grow = img_bytes > ctx->img_cap || M > ctx->M_cap || (size_t)M * L * 64 > ctx->warp_cap;
  if (grow) HIPCHK(hipStreamSynchronize(ctx->stream));                              // buffers of a frame still in flight are about to be re-allocated
  rc = ensure(ctx, ctx->d_img, ctx->img_cap, img_bytes); if (rc) return rc;
  if (M > ctx->M_cap) {
    hipError_t e;
    if (ctx->d_pos) { e = DFREE(ctx->d_pos); e = DFREE(ctx->d_invexpo); e = DFREE(ctx->d_search); e = DFREE(ctx->d_errors); (void)e; }
    const int cap = std::max(M, 512);
    HIPCHK(DMALLOC((void **)&ctx->d_pos, (size_t)cap * 24)); HIPCHK(DMALLOC((void **)&ctx->d_invexpo, (size_t)cap * 8));
    HIPCHK(DMALLOC((void **)&ctx->d_search, (size_t)cap * 4)); HIPCHK(DMALLOC((void **)&ctx->d_errors, (size_t)cap * 4));
    ctx->M_cap = cap;
  }
  rc = ensure(ctx, ctx->d_warp, ctx->warp_cap, std::max((size_t)M * L * 64, (size_t)64)); if (rc) return rc;
  {
    const int vgrid = visual_grid_inverse(std::max(M, 1));
    const size_t need = std::max(std::max((size_t)vgrid * VIS_PSTRIDE, (size_t)64), (size_t)lidar_grid(std::max(ctx->n, 1), ctx->lidar_block) * 32);      // d_partials serves both updates of the frame
    if (need > ctx->partials_cap) { HIPCHK(hipStreamSynchronize(ctx->stream)); rc = ensure(ctx, ctx->d_partials, ctx->partials_cap, need); if (rc) return rc; }
  }
  const int k = ctx->frame_stage_next; ctx->frame_stage_next ^= 1;
  if (ctx->frame_stage_used[k]) HIPCHK(hipEventSynchronize(ctx->frame_stage_ev[k]));
  if (total > ctx->frame_stage_cap[k]) {
    if (ctx->frame_stage[k]) HIPCHK(hipHostFree(ctx->frame_stage[k]));
    ctx->frame_stage[k] = nullptr; ctx->frame_stage_cap[k] = 0;
    HIPCHK(hipHostMalloc(&ctx->frame_stage[k], 2 * total));
    ctx->frame_stage_cap[k] = 2 * total;
  }
  if (!ctx->frame_stage_ev[k]) HIPCHK(hipEventCreateWithFlags(&ctx->frame_stage_ev[k], hipEventDisableTiming));
  char *h = (char *)ctx->frame_stage[k];
  std::memcpy(h, f->img, img_bytes);
  if (M > 0) {
    std::memcpy(h + o_pos, f->pos, (size_t)M * 24); std::memcpy(h + o_warp, f->warp_patch, (size_t)M * L * 256);
    std::memcpy(h + o_sl, f->search_levels, (size_t)M * 4); std::memcpy(h + o_ie, f->inv_expo_list, (size_t)M * 8);
  }
  HIPCHK(devalloc::memcpy_async(ctx->d_img, h, img_bytes, hipMemcpyHostToDevice, ctx->stream));
  if (M > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->d_pos, h + o_pos, (size_t)M * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_warp, h + o_warp, (size_t)M * L * 256, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_search, h + o_sl, (size_t)M * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->d_invexpo, h + o_ie, (size_t)M * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipEventRecord(ctx->frame_stage_ev[k], ctx->stream)); ctx->frame_stage_used[k] = true;
  ctx->width = f->width; ctx->height = f->height; ctx->stride = f->stride; ctx->M = M; ctx->L = L;
  ctx->has_frame = true; ctx->has_ref = false;
  // ---- VIO on the shared state: iterate = prior = the LiDAR posterior, handed over on the device
  if (M == 0) {                 // total_points == 0: computeJacobianAndUpdateEKF returns immediately (vio.cpp:786)
    hipLaunchKernelGGL(k_ctl_chain_visual, dim3(1), dim3(256), 0, ctx->stream, ctx->d_ctl);
    VisualKernelArgs a = make_visual_args(ctx, f->visual_cfg, 0);
    hipLaunchKernelGGL(k_visual_finish, dim3(1), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->d_ctl, a, 0);
    HIPCHK(hipGetLastError());
  } else {
    rc = visual_enqueue(ctx, nullptr, nullptr, f->visual_cfg, f->visual_cfg->patch_pyrimid_level - 1, 0, f->visual_cfg->max_iterations, 1); if (rc) return rc;
  }
  // ---- both result blocks + the watchdog flag into this frame's pinned slot
  if (!ctx->h_frame_res) HIPCHK(hipHostMalloc(&ctx->h_frame_res, 2 * sizeof(FrameRes)));
  const int slot = (ctx->frame_head + ctx->frame_inflight) & 1;
  FrameRes *r = (FrameRes *)ctx->h_frame_res + slot;
  if (!ctx->frame_res_ev[slot]) HIPCHK(hipEventCreateWithFlags(&ctx->frame_res_ev[slot], hipEventDisableTiming));
  HIPCHK(devalloc::memcpy_async(&r->lidar, &ctx->d_ctl->lidar, sizeof(livo2_lidar_result), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(&r->visual, &ctx->d_ctl->visual, sizeof(livo2_visual_result), hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(devalloc::memcpy_async(&r->timed_out, &ctx->d_ctl->hdr.pad[0], 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipEventRecord(ctx->frame_res_ev[slot], ctx->stream));
  ctx->frame_inflight++;
  return LIVO2_OK;
}

int livo2_frame_update_fetch(livo2_ctx *ctx, livo2_lidar_result *lidar, livo2_visual_result *visual) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!lidar || !visual) return fail(ctx, LIVO2_ERR_INVALID, "result is NULL");
  if (ctx->frame_inflight < 1) return fail(ctx, LIVO2_ERR_INVALID, "no frame in flight");
  HIPCHK(hipSetDevice(ctx->device));
  const int slot = ctx->frame_head;
  FrameRes *r = (FrameRes *)ctx->h_frame_res + slot;
  HIPCHK(hipEventSynchronize(ctx->frame_res_ev[slot]));
  ctx->frame_head ^= 1; ctx->frame_inflight--;
  if (r->timed_out) {
    // the resident visual grid of this frame gave up and committed nothing.  With a second frame already in flight its inputs on the device are gone: the caller
    // gets an error for THIS frame (the next one is unaffected); alone in flight, the update is re-run per step from the LiDAR posterior still in the result block
    ctx->vp_timeouts++;
    ctx->vp_backoff_len = ctx->vp_backoff_len ? std::min(2 * ctx->vp_backoff_len, 1024) : 8;
    ctx->vp_backoff_left = ctx->vp_debug_timeout ? 0 : ctx->vp_backoff_len;
    if (ctx->frame_inflight > 0) return fail(ctx, LIVO2_ERR_HIP, "the resident visual grid of this frame timed out while the next frame was already enqueued: re-submit the frame");
    HIPCHK(hipMemsetAsync(&ctx->d_ctl->hdr.pad[0], 0, 4, ctx->stream));
    const livo2_visual_cfg vc = ctx->vp_last_cfg;
    ctx->vp_rerun = true;
    const int rc = visual_enqueue(ctx, nullptr, nullptr, &vc, vc.patch_pyrimid_level - 1, 0, vc.max_iterations, 1);
    ctx->vp_rerun = false;
    if (rc) return rc;
    HIPCHK(devalloc::memcpy_async(&r->visual, &ctx->d_ctl->visual, sizeof(livo2_visual_result), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
  } else if (ctx->vp_backoff_left == 0) ctx->vp_backoff_len = 0;
  std::memcpy(lidar, &r->lidar, sizeof(livo2_lidar_result));
  std::memcpy(visual, &r->visual, sizeof(livo2_visual_result));
  return rz_gate(ctx);
}

int livo2_frame_update(livo2_ctx *ctx, const livo2_frame_in *frame, livo2_lidar_result *lidar, livo2_visual_result *visual) {
  int rc = livo2_frame_update_async(ctx, frame); if (rc) return rc;
  return livo2_frame_update_fetch(ctx, lidar, visual);
}

int livo2_visual_iterations_async(livo2_ctx *ctx, int32_t level, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg,
                                  int32_t iters) {
  int rc = visual_ready(ctx, state_in, prop, cfg); if (rc) return rc;
  if (iters < 1 || level < 0 || level >= cfg->patch_pyrimid_level) return fail(ctx, LIVO2_ERR_INVALID, "bad iters/level");
  if (ctx->M == 0) return fail(ctx, LIVO2_ERR_INVALID, "no patches");
  return visual_enqueue(ctx, state_in, prop, cfg, level, level, iters, 2);
}

// ---- batch of frames, visual ------------------------------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(LIVO2_WAVE) k_visual_finish_batch(const VisualBatchEntry *__restrict__ entries, int update_cov) {
  const VisualBatchEntry &e = entries[blockIdx.x];
  visual_finish_body(e.ctl, e.a, (update_cov && e.a.M > 0) ? 1 : 0);      // total_points == 0: no update, the covariance stays (vio.cpp:786)
}
__global__ void __launch_bounds__(256) k_vbatch_gather_out(const VisualBatchEntry *__restrict__ entries, livo2_visual_result *__restrict__ out) {
  const double *src = reinterpret_cast<const double *>(&entries[blockIdx.x].ctl->visual);
  double *dst = reinterpret_cast<double *>(out + blockIdx.x);
  for (int e = threadIdx.x; e < (int)(sizeof(livo2_visual_result) / sizeof(double)); e += 256) dst[e] = src[e];
}
static_assert(sizeof(livo2_visual_result) % 8 == 0, "copied as doubles");
} // namespace

int livo2_visual_batch_set_frames(livo2_ctx *ctx, int32_t n_frames, const uint8_t *imgs, int32_t width, int32_t height, int32_t stride, const double *pos,
                                  const float *warp_patch, const int32_t *search_levels, const double *inv_expo_list, const int32_t *counts, int32_t L) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (n_frames < 1 || n_frames > LIVO2_MAX_BATCH) return fail(ctx, LIVO2_ERR_INVALID, "n_frames out of [1,LIVO2_MAX_BATCH]");
  if (!imgs || width <= 0 || height <= 0 || stride < width) return fail(ctx, LIVO2_ERR_INVALID, "bad image");
  if (!counts || L < 1 || L > LIVO2_MAX_LEVELS) return fail(ctx, LIVO2_ERR_INVALID, "bad counts / L");
  long long total = 0;
  for (int f = 0; f < n_frames; f++) { if (counts[f] < 0) return fail(ctx, LIVO2_ERR_INVALID, "negative patch count"); total += counts[f]; }
  if (total > (1ll << 26)) return fail(ctx, LIVO2_ERR_INVALID, "batch too large");
  if (total > 0 && (!pos || !warp_patch || !search_levels || !inv_expo_list)) return fail(ctx, LIVO2_ERR_INVALID, "bad sub-map arrays");
  for (long long i = 0; i < total; i++) if (search_levels[i] < 0 || search_levels[i] > 8) return fail(ctx, LIVO2_ERR_RANGE, "search_level out of [0,8]");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  int rc = batch_alloc_fixed(ctx); if (rc) return rc;
  if (!ctx->vbd_entries) {
    HIPCHK(DMALLOC((void **)&ctx->vbd_entries, sizeof(VisualBatchEntry) * LIVO2_MAX_BATCH));
    HIPCHK(DMALLOC((void **)&ctx->vbd_results, sizeof(livo2_visual_result) * LIVO2_MAX_BATCH));
    HIPCHK(hipHostMalloc((void **)&ctx->vbh_entries, sizeof(VisualBatchEntry) * LIVO2_MAX_BATCH));
    HIPCHK(hipHostMalloc((void **)&ctx->vbh_results, sizeof(livo2_visual_result) * LIVO2_MAX_BATCH));
  }
  const size_t img_bytes = (size_t)stride * height, T = (size_t)std::max<long long>(total, 1);
  if ((rc = ensure(ctx, ctx->vbd_img, ctx->vb_img_cap, img_bytes * n_frames))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_pos, ctx->vb_pos_cap, T * 3))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_invexpo, ctx->vb_invexpo_cap, T))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_search, ctx->vb_search_cap, T))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_errors, ctx->vb_errors_cap, T))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_warp, ctx->vb_warp_cap, T * L * 64))) return rc;
  ctx->vbn = n_frames; ctx->vb_total = (int)total; ctx->vb_L = L; ctx->vb_w = width; ctx->vb_h = height; ctx->vb_stride = stride;
  ctx->vb_count.assign(counts, counts + n_frames);
  ctx->vb_off.resize(n_frames); ctx->vb_grid.resize(n_frames); ctx->vb_block_begin.resize(n_frames);
  int off = 0, blocks = 0;
  for (int f = 0; f < n_frames; f++) { ctx->vb_off[f] = off; ctx->vb_grid[f] = visual_grid(std::max(counts[f], 1)); ctx->vb_block_begin[f] = blocks; off += counts[f]; blocks += ctx->vb_grid[f]; }
  ctx->vb_blocks = blocks;
  if ((rc = ensure(ctx, ctx->vbd_partials, ctx->vb_partials_cap, (size_t)blocks * VIS_PSTRIDE))) return rc;
  if ((rc = ensure(ctx, ctx->vbd_block_frame, ctx->vb_block_frame_cap, (size_t)blocks))) return rc;
  {
    std::vector<int32_t> bf((size_t)blocks);
    for (int f = 0; f < n_frames; f++) std::fill(bf.begin() + ctx->vb_block_begin[f], bf.begin() + ctx->vb_block_begin[f] + ctx->vb_grid[f], f);
    HIPCHK(hipMemcpy(ctx->vbd_block_frame, bf.data(), (size_t)blocks * 4, hipMemcpyHostToDevice));
  }
  HIPCHK(devalloc::memcpy_async(ctx->vbd_img, imgs, img_bytes * n_frames, hipMemcpyHostToDevice, ctx->stream));
  if (total > 0) {
    HIPCHK(devalloc::memcpy_async(ctx->vbd_pos, pos, (size_t)total * 24, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->vbd_warp, warp_patch, (size_t)total * L * 256, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->vbd_search, search_levels, (size_t)total * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(devalloc::memcpy_async(ctx->vbd_invexpo, inv_expo_list, (size_t)total * 8, hipMemcpyHostToDevice, ctx->stream));
  }
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->has_vbatch = true;
  return LIVO2_OK;
}

static int vbatch_enqueue(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg, int level_hi, int level_lo,
                          int iters, int mode) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!state_in || !prop || !cfg) return fail(ctx, LIVO2_ERR_INVALID, "state / cfg is NULL");
  if (!ctx->has_vbatch) return fail(ctx, LIVO2_ERR_NO_FRAME, "livo2_visual_batch_set_frames has not been called");
  if (n_frames != ctx->vbn) return fail(ctx, LIVO2_ERR_INVALID, "n_frames differs from the batch set by livo2_visual_batch_set_frames");
  if (cfg->inverse_composition_en) return fail(ctx, LIVO2_ERR_INVALID, "the batched visual update is forward-compositional only");
  if (cfg->max_iterations < 1 || cfg->max_iterations > LIVO2_MAX_ITERS || cfg->patch_pyrimid_level < 1 || cfg->patch_pyrimid_level > ctx->vb_L || !(cfg->img_point_cov > 0) ||
      cfg->mp_proc_num < 0 || cfg->mp_proc_num > LIVO2_WAVE)
    return fail(ctx, LIVO2_ERR_INVALID, "bad visual cfg");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipStreamSynchronize(ctx->stream));        // pinned staging blocks are reused
  // per-frame kernel arguments: the camera / extrinsic constants of make_visual_args on this frame's slices
  const int keepM = ctx->M, keepL = ctx->L;
  for (int f = 0; f < n_frames; f++) {
    HostIn &h = ctx->bh_in[f];
    h.cur = state_in[f]; h.prop = prop[f];
    std::memset(&h.hdr, 0, sizeof(DevHeader));
    h.hdr.last_error = FLT_MAX;
    VisualBatchEntry &e = ctx->vbh_entries[f];
    ctx->M = ctx->vb_count[f]; ctx->L = ctx->vb_L;
    e.a = make_visual_args(ctx, cfg, 0);
    const size_t o = (size_t)ctx->vb_off[f];
    e.a.img = ctx->vbd_img + (size_t)f * ctx->vb_stride * ctx->vb_h; e.a.width = ctx->vb_w; e.a.height = ctx->vb_h; e.a.stride = ctx->vb_stride;
    e.a.pos = ctx->vbd_pos + o * 3; e.a.warp = ctx->vbd_warp + o * ctx->vb_L * 64; e.a.search_levels = ctx->vbd_search + o; e.a.inv_expo = ctx->vbd_invexpo + o;
    e.a.errors = ctx->vbd_errors + o; e.a.z = nullptr; e.a.H_sub = nullptr;
    e.ctl = ctx->bd_ctl + f; e.partials = ctx->vbd_partials + (size_t)ctx->vb_block_begin[f] * VIS_PSTRIDE; e.block_begin = ctx->vb_block_begin[f]; e.nblocks = ctx->vb_grid[f];
  }
  ctx->M = keepM; ctx->L = keepL;
  HIPCHK(devalloc::memcpy_async(ctx->bd_in, ctx->bh_in, sizeof(HostIn) * n_frames, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(ctx->vbd_entries, ctx->vbh_entries, sizeof(VisualBatchEntry) * n_frames, hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_batch_scatter_in, dim3(n_frames), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->bd_in, ctx->bd_ctl);
  for (int level = level_hi; level >= level_lo; level--)
    for (int it = 0; it < iters; it++) {
      { Timed t(ctx, 1); hipLaunchKernelGGL(k_visual_residual_batch, dim3(ctx->vb_blocks), dim3(VIS_BLOCK), 0, ctx->stream, ctx->vbd_entries, ctx->vbd_block_frame, level, (mode == 1 && it > 0) ? 1 : 0); t.done(); }
      { Timed t(ctx, 3); hipLaunchKernelGGL(k_visual_solve_batch, dim3(n_frames), dim3(512), 0, ctx->stream, ctx->vbd_entries, mode, level, mode == 1 ? it : (it == 0 ? 0 : 1), cfg->img_point_cov, cfg->mp_proc_num); t.done(); }
    }
  hipLaunchKernelGGL(k_visual_finish_batch, dim3(n_frames), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->vbd_entries, mode == 1 ? 1 : 0);
  hipLaunchKernelGGL(k_vbatch_gather_out, dim3(n_frames), dim3(256), 0, ctx->stream, ctx->vbd_entries, ctx->vbd_results);
  HIPCHK(hipGetLastError());
  return LIVO2_OK;
}

int livo2_visual_batch_update_async(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg) {
  if (!ctx || !cfg) return ctx ? fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL") : LIVO2_ERR_INVALID;
  return vbatch_enqueue(ctx, n_frames, state_in, prop, cfg, cfg->patch_pyrimid_level - 1, 0, cfg->max_iterations, 1);
}
int livo2_visual_batch_update_fetch(livo2_ctx *ctx, int32_t n_frames, livo2_visual_result *results) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!results) return fail(ctx, LIVO2_ERR_INVALID, "results is NULL");
  if (!ctx->has_vbatch || n_frames != ctx->vbn) return fail(ctx, LIVO2_ERR_INVALID, "n_frames differs from the batch set by livo2_visual_batch_set_frames");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(devalloc::memcpy_async(ctx->vbh_results, ctx->vbd_results, sizeof(livo2_visual_result) * n_frames, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  std::memcpy(results, ctx->vbh_results, sizeof(livo2_visual_result) * n_frames);
  return rz_gate(ctx);
}
int livo2_visual_batch_update(livo2_ctx *ctx, int32_t n_frames, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg, livo2_visual_result *results) {
  int rc = livo2_visual_batch_update_async(ctx, n_frames, state_in, prop, cfg); if (rc) return rc;
  return livo2_visual_batch_update_fetch(ctx, n_frames, results);
}
int livo2_visual_batch_iterations_async(livo2_ctx *ctx, int32_t n_frames, int32_t level, const livo2_state *state_in, const livo2_state *prop, const livo2_visual_cfg *cfg,
                                        int32_t iters) {
  if (!ctx || !cfg) return ctx ? fail(ctx, LIVO2_ERR_INVALID, "cfg is NULL") : LIVO2_ERR_INVALID;
  if (iters < 1 || level < 0 || level >= cfg->patch_pyrimid_level) return fail(ctx, LIVO2_ERR_INVALID, "bad iters/level");
  return vbatch_enqueue(ctx, n_frames, state_in, prop, cfg, level, level, iters, 2);
}

// LIVO2_VP_PROF=1: stamps of the last persistent visual update, [VP_MAX_BLOCKS = 256 blocks][32 steps][16] (tools/vis_persist_probe.py)
int livo2_debug_vp_prof(livo2_ctx *ctx, unsigned long long *out) {
  if (!ctx || !out || !ctx->d_vp_prof) return LIVO2_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, ctx->d_vp_prof, VP_MAX_BLOCKS * 32 * 16 * 8, hipMemcpyDeviceToHost));
  return LIVO2_OK;
}
#ifdef LIVO2_PHASE_PROF
// profiling build only: per-wave stamps of the LAST k_visual_residual launch, [waves][8] (tools/vis_phase.py)
int livo2_debug_vis_prof(livo2_ctx *ctx, unsigned long long *out, size_t n_waves) {
  if (!ctx || !out) return LIVO2_ERR_INVALID;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vis_prof), std::min(n_waves, (size_t)VIS_PROF_WAVES) * 64));
  return LIVO2_OK;
}
// profiling build only: copy the per-wave phase stamps of the LAST residual launch to host memory
int livo2_debug_phase_prof(livo2_ctx *ctx, unsigned long long *out, size_t max_waves, size_t *n_waves) {
  if (!ctx || !ctx->d_prof) return LIVO2_ERR_INVALID;
  // rows [0, waves): per-wave phase stamps; rows waves, waves+1: solve-kernel stamps; then waves x 2 chip-wide clock stamps
  size_t w = std::min(max_waves, ctx->prof_waves + 2 + (ctx->prof_waves + 3) / 4 + 2 * ctx->prof_waves);
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, ctx->d_prof, std::min(w * 64, ctx->prof_waves * 208 + 128), hipMemcpyDeviceToHost));
  if (n_waves) *n_waves = w;
  return LIVO2_OK;
}
#endif

// ---- solve alone ----------------------------------------------------------------------------------------------------------------
int livo2_esikf_solve(livo2_ctx *ctx, const double *HtH, const double *Htz, int32_t k, double meas_cov_scale, int32_t sign, const livo2_state *cur,
                      const livo2_state *prop, livo2_state *out_state, double *solution, double *G) {
  if (!ctx) return LIVO2_ERR_INVALID;
  if (!HtH || !Htz || !cur || !prop || (k != 6 && k != 7) || !(meas_cov_scale > 0) || (sign != 1 && sign != -1)) return fail(ctx, LIVO2_ERR_INVALID, "bad arguments");
  HIPCHK(hipSetDevice(ctx->device));
  int rc = upload_states(ctx, cur, prop); if (rc) return rc;
  HIPCHK(devalloc::memcpy_async(ctx->d_ctl->solve_hth, HtH, (size_t)k * k * 8, hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(devalloc::memcpy_async(ctx->d_ctl->solve_htz, Htz, (size_t)k * 8, hipMemcpyHostToDevice, ctx->stream));
  { Timed t(ctx, 2); hipLaunchKernelGGL(k_esikf_solve_only, dim3(1), dim3(LIVO2_WAVE), 0, ctx->stream, ctx->d_ctl, k, meas_cov_scale, sign); t.done(); }
  HIPCHK(hipGetLastError());
  if (out_state) HIPCHK(devalloc::memcpy_async(out_state, &ctx->d_ctl->cur, sizeof(livo2_state), hipMemcpyDeviceToHost, ctx->stream));
  if (solution) HIPCHK(devalloc::memcpy_async(solution, ctx->d_ctl->solve_solution, DS * 8, hipMemcpyDeviceToHost, ctx->stream));
  if (G) HIPCHK(devalloc::memcpy_async(G, ctx->d_ctl->G, DS * DS * 8, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return LIVO2_OK;
}

} // extern "C"
