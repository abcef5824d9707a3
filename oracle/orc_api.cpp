// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
//
// PARITY UNPINNED.  The reference ships no golden vectors, known-answer tests or fixtures for this path, and it cannot
// be built in this image: its translation units need ROS1/catkin, PCL, OpenCV, Eigen3, Sophus and rpg_vikit, none of which are present
// (and a build against written stand-ins for them is not a reference build).  This restatement therefore stands on the cited reference
// lines, on the numpy second opinions and hand-computed known answers in tests/, and on nothing stronger.
//
// extern "C" surface over the restated reference path (orc_lidar.hpp / orc_visual.hpp / orc_voxel_map.hpp) so that the
// Python test-suite can drive it through ctypes.  "Flat map" = the neutral array interchange format both the oracle and
// the product's C-ABI (include/livo2_hip.h) can be fed from:
//   roots : key int64[R][3], node int32[R], center f64[R][3], quarter f32[R]
//   nodes : plane int32[Nn] (-1 when !is_plane_), child int32[Nn][8] (-1 when null)
//   planes: normal f64[P][3], center f64[P][3], plane_var f64[P][36], d f32[P], radius f32[P]
#include "orc_lidar.hpp"
#include "orc_visual.hpp"
#include "orc_warp.hpp"
#include "orc_preprocess.hpp"
#include "orc_select.hpp"
#include "orc_choice.hpp"
#include "orc_imu.hpp"
#include <chrono>

using namespace orc;

namespace {
struct MapHandle {
  VoxelMap map;
  std::unordered_map<const VoxelPlane *, int> plane_index;   // filled by export / from_flat
  std::vector<VoxelOctoTree *> root_order;                   // export order of roots
  std::vector<VOXEL_LOCATION> root_keys;
};

void count_nodes(const VoxelOctoTree *t, int &n_nodes, int &n_planes) {
  n_nodes++;
  if (t->plane_ptr_->is_plane_) n_planes++;
  for (int i = 0; i < 8; i++) if (t->leaves_[i]) count_nodes(t->leaves_[i], n_nodes, n_planes);
}

struct Exporter {
  MapHandle *h; int32_t *node_plane, *node_child; double *pn, *pc, *pv; float *pd, *pr; int n_nodes = 0, n_planes = 0;
  int visit(const VoxelOctoTree *t) {
    int me = n_nodes++;
    if (t->plane_ptr_->is_plane_) {
      const VoxelPlane *p = t->plane_ptr_;
      int pi = n_planes++;
      h->plane_index[p] = pi;
      for (int k = 0; k < 3; k++) { pn[pi * 3 + k] = p->normal_[k]; pc[pi * 3 + k] = p->center_[k]; }
      for (int k = 0; k < 36; k++) pv[(size_t)pi * 36 + k] = p->plane_var_.a[k];
      pd[pi] = p->d_; pr[pi] = p->radius_;
      node_plane[me] = pi;
    } else node_plane[me] = -1;
    for (int i = 0; i < 8; i++) node_child[(size_t)me * 8 + i] = -1;
    for (int i = 0; i < 8; i++) if (t->leaves_[i]) { int c = visit(t->leaves_[i]); node_child[(size_t)me * 8 + i] = c; }
    return me;
  }
};
} // namespace

extern "C" {

// summation-order model of the long reductions (orc_math.hpp): 0 = strict left-to-right (parity checker), 1 = Eigen-like (sensitivity study only)
void orc_set_sum_model(int mode, int kc, int lanes) { orc::sum_model().mode = mode; orc::sum_model().kc = kc > 0 ? kc : 256; orc::sum_model().lanes = (lanes == 2 || lanes == 4 || lanes == 8) ? lanes : 4; }

struct orc_lidar_cfg {
  int max_iterations; int max_layer;
  double sigma_num, dept_err, beam_err, voxel_size, deg2rad;
  double extR[9], extT[3];
  int num_threads; int pad;
};

struct orc_visual_cfg {
  double fx, fy, cx, cy, d[5];
  int distortion, width, height, patch_pyrimid_level, max_iterations, exposure_estimate_en, inverse_composition_en, num_threads;
  double img_point_cov;
  double Rcl[9], Pcl[3], extR[9], extT[3];
};

int orc_sizeof_lidar_trace() { return (int)sizeof(LidarIterTrace); }
int orc_sizeof_visual_trace() { return (int)sizeof(VisualIterTrace); }
int orc_sizeof_state() { return (int)sizeof(StatePOD); }

void *orc_map_create(double voxel_size, int max_layer, const int *layer_init_num5, int max_points_num, double planer_threshold) {
  MapHandle *h = new MapHandle;
  h->map.config_setting_.max_voxel_size_ = voxel_size;
  h->map.config_setting_.max_layer_ = max_layer;
  h->map.config_setting_.layer_init_num_.assign(layer_init_num5, layer_init_num5 + 5);
  h->map.config_setting_.max_points_num_ = max_points_num;
  h->map.config_setting_.planner_threshold_ = planer_threshold;
  return h;
}
void orc_map_destroy(void *m) { delete (MapHandle *)m; }

static void to_points(const double *pw, const double *var9, int n, std::vector<MapPoint> &pts) {
  pts.resize(n);
  for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) pts[i].point_w[k] = pw[(size_t)i * 3 + k]; for (int k = 0; k < 9; k++) pts[i].var.a[k] = var9[(size_t)i * 9 + k]; }
}
void orc_map_build(void *m, const double *pw, const double *var9, int n) { std::vector<MapPoint> pts; to_points(pw, var9, n, pts); ((MapHandle *)m)->map.BuildVoxelMap(pts); }
void orc_map_update(void *m, const double *pw, const double *var9, int n) { std::vector<MapPoint> pts; to_points(pw, var9, n, pts); ((MapHandle *)m)->map.UpdateVoxelMap(pts); }

int orc_map_slide(void *m, const double *position_last, double sliding_thresh, int half_map_size) {
  return ((MapHandle *)m)->map.mapSliding(vec3(position_last[0], position_last[1], position_last[2]), sliding_thresh, half_map_size);
}
void orc_map_counts(void *m, int *n_roots, int *n_nodes, int *n_planes) {
  MapHandle *h = (MapHandle *)m; int nn = 0, np = 0;
  for (auto &kv : h->map.voxel_map_) count_nodes(kv.second, nn, np);
  *n_roots = (int)h->map.voxel_map_.size(); *n_nodes = nn; *n_planes = np;
}

void orc_map_export(void *m, int64_t *keys, int32_t *root_node, double *root_center, float *root_quarter, int32_t *node_plane, int32_t *node_child,
                    double *plane_normal, double *plane_center, double *plane_var, float *plane_d, float *plane_radius) {
  MapHandle *h = (MapHandle *)m;
  h->plane_index.clear(); h->root_order.clear(); h->root_keys.clear();
  Exporter ex{h, node_plane, node_child, plane_normal, plane_center, plane_var, plane_d, plane_radius};
  int r = 0;
  for (auto &kv : h->map.voxel_map_) {
    keys[r * 3 + 0] = kv.first.x; keys[r * 3 + 1] = kv.first.y; keys[r * 3 + 2] = kv.first.z;
    for (int k = 0; k < 3; k++) root_center[r * 3 + k] = kv.second->voxel_center_[k];
    root_quarter[r] = kv.second->quater_length_;
    root_node[r] = ex.visit(kv.second);
    h->root_order.push_back(kv.second); h->root_keys.push_back(kv.first);
    r++;
  }
}

static VoxelOctoTree *build_from_flat(MapHandle *h, int node, int layer, const int32_t *node_plane, const int32_t *node_child, const double *pn,
                                      const double *pc, const double *pv, const float *pd, const float *pr) {
  auto &cfg = h->map.config_setting_;
  VoxelOctoTree *t = new VoxelOctoTree(cfg.max_layer_, layer, 5, cfg.max_points_num_, (float)cfg.planner_threshold_, &h->map.voxel_plane_id);
  t->layer_init_num_ = cfg.layer_init_num_;
  t->init_octo_ = true;
  int pi = node_plane[node];
  if (pi >= 0) {
    VoxelPlane *p = t->plane_ptr_;
    for (int k = 0; k < 3; k++) { p->normal_[k] = pn[pi * 3 + k]; p->center_[k] = pc[pi * 3 + k]; }
    for (int k = 0; k < 36; k++) p->plane_var_.a[k] = pv[(size_t)pi * 36 + k];
    p->d_ = pd[pi]; p->radius_ = pr[pi]; p->is_plane_ = true; p->is_init_ = true;
    h->plane_index[p] = pi;
  }
  for (int i = 0; i < 8; i++) { int c = node_child[(size_t)node * 8 + i]; if (c >= 0) t->leaves_[i] = build_from_flat(h, c, layer + 1, node_plane, node_child, pn, pc, pv, pd, pr); }
  return t;
}

void *orc_map_from_flat(double voxel_size, int max_layer, int n_roots, const int64_t *keys, const int32_t *root_node, const double *root_center,
                        const float *root_quarter, const int32_t *node_plane, const int32_t *node_child, const double *plane_normal,
                        const double *plane_center, const double *plane_var, const float *plane_d, const float *plane_radius) {
  int lin[5] = {5, 5, 5, 5, 5};
  MapHandle *h = (MapHandle *)orc_map_create(voxel_size, max_layer, lin, 50, 0.0025);
  for (int r = 0; r < n_roots; r++) {
    VoxelOctoTree *t = build_from_flat(h, root_node[r], 0, node_plane, node_child, plane_normal, plane_center, plane_var, plane_d, plane_radius);
    for (int k = 0; k < 3; k++) t->voxel_center_[k] = root_center[r * 3 + k];
    t->quater_length_ = root_quarter[r];
    h->map.voxel_map_[VOXEL_LOCATION(keys[r * 3], keys[r * 3 + 1], keys[r * 3 + 2])] = t;
  }
  return h;
}

static void setup_manager(VoxelMapManager &mgr, MapHandle *h, const orc_lidar_cfg *cfg, const float *xyz, int n, const StatePOD *state_in) {
  mgr.map_ = &h->map;
  mgr.config_setting_ = h->map.config_setting_;
  mgr.config_setting_.max_iterations_ = cfg->max_iterations; mgr.config_setting_.max_layer_ = cfg->max_layer;
  mgr.config_setting_.sigma_num_ = cfg->sigma_num; mgr.config_setting_.dept_err_ = cfg->dept_err; mgr.config_setting_.beam_err_ = cfg->beam_err;
  mgr.config_setting_.max_voxel_size_ = cfg->voxel_size;
  mgr.deg2rad_ = cfg->deg2rad; mgr.num_threads_ = cfg->num_threads > 0 ? cfg->num_threads : 1;
  std::memcpy(mgr.extR_.a, cfg->extR, 72); std::memcpy(mgr.extT_.a, cfg->extT, 24);
  mgr.feats_down_body_.resize(n);
  for (int i = 0; i < n; i++) { PointXYZINormal p; std::memset(&p, 0, sizeof(p)); p.x = xyz[(size_t)i * 3]; p.y = xyz[(size_t)i * 3 + 1]; p.z = xyz[(size_t)i * 3 + 2]; mgr.feats_down_body_[i] = p; }
  mgr.feats_down_size_ = n;
  mgr.state_.from_pod(*state_in);
}

static void dump_points(VoxelMapManager &mgr, MapHandle *h, int n, int32_t *match_plane, float *dis, float *pw, double *var, double *body_cov,
                        double *cross_mat, double *normal, double *Rinv, double *Hrow) {
  if (match_plane) for (int i = 0; i < n; i++) match_plane[i] = -1;
  if (dis) for (int i = 0; i < n; i++) dis[i] = 0;
  if (Rinv) for (int i = 0; i < n; i++) Rinv[i] = 0;
  if (Hrow) for (size_t i = 0; i < (size_t)n * 6; i++) Hrow[i] = 0;
  for (int i = 0; i < n; i++) {
    const pointWithVar &pv = mgr.pv_list_[i];
    if (pw) for (int k = 0; k < 3; k++) pw[(size_t)i * 3 + k] = (float)pv.point_w[k];
    if (var) std::memcpy(var + (size_t)i * 9, pv.var.a, 72);
    if (body_cov) std::memcpy(body_cov + (size_t)i * 9, mgr.body_cov_list_[i].a, 72);
    if (cross_mat) std::memcpy(cross_mat + (size_t)i * 9, mgr.cross_mat_list_[i].a, 72);
    if (normal) std::memcpy(normal + (size_t)i * 3, pv.normal.a, 24);
  }
  for (size_t j = 0; j < mgr.ptpl_list_.size(); j++) {
    int i = mgr.ptpl_index_[j];
    if (match_plane) { auto it = h->plane_index.find(mgr.ptpl_list_[j].plane_src_); match_plane[i] = (it == h->plane_index.end()) ? -2 : it->second; }
    if (dis) dis[i] = mgr.ptpl_list_[j].dis_to_plane_;
    if (Rinv && j < mgr.dump_Rinv_.size()) Rinv[i] = mgr.dump_Rinv_[j];
    if (Hrow && (j + 1) * 6 <= mgr.dump_H_.size()) std::memcpy(Hrow + (size_t)i * 6, &mgr.dump_H_[j * 6], 48);
  }
}

// One iteration of residual+Jacobian+reduction for a given current/prior state (no solve).
int orc_lidar_iterate(void *m, const orc_lidar_cfg *cfg, const float *xyz, int n, const StatePOD *state_cur, const StatePOD *state_prop, double *HtH36,
                      double *Htz6, int *n_eff, double *total_residual, int32_t *match_plane, float *dis, float *pw, double *var, double *body_cov,
                      double *cross_mat, double *normal, double *Rinv, double *Hrow) {
  MapHandle *h = (MapHandle *)m;
  VoxelMapManager mgr; setup_manager(mgr, h, cfg, xyz, n, state_cur);
  StatesGroup prop; prop.from_pod(*state_prop);
  mgr.per_scan_precompute();
  double tr;
  mgr.iterate_residual_and_reduce(prop, HtH36, Htz6, tr);
  *n_eff = mgr.effct_feat_num_; *total_residual = tr;
  dump_points(mgr, h, n, match_plane, dis, pw, var, body_cov, cross_mat, normal, Rinv, Hrow);
  return 0;
}

// Full StateEstimation (src/voxel_map.cpp:338-511).  `seconds` = wall time around StateEstimation only (the reference's "ICP" window,
// src/LIVMapper.cpp:368-374).
int orc_lidar_state_estimation(void *m, const orc_lidar_cfg *cfg, const float *xyz, int n, const StatePOD *state_in, const StatePOD *state_prop,
                               StatePOD *state_out, void *trace /*LidarIterTrace[max_iterations]*/, int *n_iters, double *seconds, int32_t *match_plane,
                               float *dis, float *pw, double *var, double *body_cov, double *cross_mat, double *normal, double *Rinv, double *Hrow) {
  MapHandle *h = (MapHandle *)m;
  VoxelMapManager mgr; setup_manager(mgr, h, cfg, xyz, n, state_in);
  StatesGroup prop; prop.from_pod(*state_prop);
  auto t0 = std::chrono::steady_clock::now();
  mgr.StateEstimation(prop);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  mgr.state_.to_pod(*state_out);
  if (n_iters) *n_iters = (int)mgr.trace_.size();
  if (trace) std::memcpy(trace, mgr.trace_.data(), mgr.trace_.size() * sizeof(LidarIterTrace));
  dump_points(mgr, h, n, match_plane, dis, pw, var, body_cov, cross_mat, normal, Rinv, Hrow);
  return 0;
}

static void setup_vio(VIOManager &vio, SubSparseMap &sm, std::vector<VisualPoint> &pts, const orc_visual_cfg *cfg, const double *pos, const float *warp_patch,
                      const int32_t *search_levels, const double *inv_expo_list, int M, const uint8_t *ref_imgs, const int32_t *ref_img_idx,
                      const double *ref_px, const double *ref_f, const double *ref_R, const double *ref_pos) {
  vio.cam.fx = cfg->fx; vio.cam.fy = cfg->fy; vio.cam.cx = cfg->cx; vio.cam.cy = cfg->cy; std::memcpy(vio.cam.d, cfg->d, 40);
  vio.cam.distortion = cfg->distortion; vio.cam.width = cfg->width; vio.cam.height = cfg->height;
  M3 extR, Rcl; V3 extT, Pcl;
  std::memcpy(extR.a, cfg->extR, 72); std::memcpy(extT.a, cfg->extT, 24); std::memcpy(Rcl.a, cfg->Rcl, 72); std::memcpy(Pcl.a, cfg->Pcl, 24);
  vio.patch_size = 8; vio.patch_pyrimid_level = cfg->patch_pyrimid_level; vio.max_iterations = cfg->max_iterations;
  vio.img_point_cov = cfg->img_point_cov; vio.exposure_estimate_en = cfg->exposure_estimate_en; vio.inverse_composition_en = cfg->inverse_composition_en;
  vio.num_threads_ = cfg->num_threads > 0 ? cfg->num_threads : 1;
  vio.setImuToLidarExtrinsic(extT, extR);       // LIVMapper.cpp:133
  vio.setLidarToCameraExtrinsic(Rcl, Pcl);      // LIVMapper.cpp:134
  vio.initializeVIO();
  const int L = cfg->patch_pyrimid_level;
  pts.resize(M);
  sm.errors.assign(M, 0.f); sm.warp_patch.resize(M); sm.search_levels.resize(M); sm.voxel_points.resize(M); sm.inv_expo_list.resize(M);
  for (int i = 0; i < M; i++) {
    for (int k = 0; k < 3; k++) pts[i].pos_[k] = pos[(size_t)i * 3 + k];
    sm.voxel_points[i] = &pts[i];
    sm.warp_patch[i].assign(warp_patch + (size_t)i * L * 64, warp_patch + (size_t)(i + 1) * L * 64);
    sm.search_levels[i] = search_levels[i];
    sm.inv_expo_list[i] = inv_expo_list[i];
  }
  if (ref_imgs && ref_img_idx) {
    sm.ref_patches.resize(M);
    for (int i = 0; i < M; i++) {
      RefPatch &rp = sm.ref_patches[i];
      rp.img_ = ref_imgs + (size_t)ref_img_idx[i] * cfg->width * cfg->height;
      rp.px_[0] = ref_px[i * 2]; rp.px_[1] = ref_px[i * 2 + 1];
      for (int k = 0; k < 3; k++) { rp.f_[k] = ref_f[i * 3 + k]; rp.pos_ref[k] = ref_pos[i * 3 + k]; }
      std::memcpy(rp.R_ref_w.a, ref_R + (size_t)i * 9, 72);
    }
  }
  vio.visual_submap = &sm; vio.total_points = M;
}

// One forward-compositional evaluation at `level` for a given state: z, H_sub (H_DIM x 7), per-patch errors, 7x7 / 7x1 sums.
int orc_visual_iterate(const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels,
                       const double *inv_expo_list, int M, int level, const StatePOD *state_cur, double *z, double *H_sub, float *errors, double *HtH49,
                       double *Htz7, float *error, int *n_meas) {
  VIOManager vio; SubSparseMap sm; std::vector<VisualPoint> pts;
  setup_vio(vio, sm, pts, cfg, pos, warp_patch, search_levels, inv_expo_list, M, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  StatesGroup st; st.from_pod(*state_cur); vio.state = &st; vio.state_propagat = &st;
  const int H_DIM = M * 64;
  std::vector<double> zz(H_DIM, 0.0), HH((size_t)H_DIM * 7, 0.0);
  int nm = 0;
  float e = vio.eval_forward(img, level, zz, HH, nm);
  if (z) std::memcpy(z, zz.data(), zz.size() * 8);
  if (H_sub) std::memcpy(H_sub, HH.data(), HH.size() * 8);
  if (errors) std::memcpy(errors, sm.errors.data(), M * 4);
  for (int a = 0; a < 7; a++) {
    double s = 0.0; for (int r = 0; r < H_DIM; r++) s += HH[(size_t)r * 7 + a] * zz[r];
    if (Htz7) Htz7[a] = s;
    for (int b = 0; b < 7; b++) { double t = 0.0; for (int r = 0; r < H_DIM; r++) t += HH[(size_t)r * 7 + a] * HH[(size_t)r * 7 + b]; if (HtH49) HtH49[a * 7 + b] = t; }
  }
  *error = e; *n_meas = nm;
  return 0;
}

// One inverse-compositional evaluation at `level`: precomputeReferencePatches(level) + the residual / Jacobian part of updateStateInverse
// (src/vio.cpp:1327-1477) for the given state: z, H_sub (H_DIM x 6), errors, 6x6 / 6x1 sums, mean error.
int orc_visual_iterate_inverse(const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels,
                               const double *inv_expo_list, int M, int level, const StatePOD *state_cur, const uint8_t *ref_imgs, const int32_t *ref_img_idx,
                               const double *ref_px, const double *ref_f, const double *ref_R, const double *ref_pos, double *z, double *H_sub6, float *errors,
                               double *HtH36, double *Htz6, float *error, int *n_meas) {
  VIOManager vio; SubSparseMap sm; std::vector<VisualPoint> pts;
  setup_vio(vio, sm, pts, cfg, pos, warp_patch, search_levels, inv_expo_list, M, ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos);
  StatesGroup st, prop; st.from_pod(*state_cur); prop.from_pod(*state_cur); vio.state = &st; vio.state_propagat = &prop;
  vio.max_iterations = 1; vio.has_ref_patch_cache = false; vio.trace_.clear();
  vio.updateStateInverse(img, level);
  if (z) std::memcpy(z, vio.dump_z_.data(), vio.dump_z_.size() * 8);
  if (H_sub6) std::memcpy(H_sub6, vio.dump_H_.data(), vio.dump_H_.size() * 8);
  if (errors) std::memcpy(errors, sm.errors.data(), M * 4);
  const VisualIterTrace &tr = vio.trace_.at(0);
  for (int a = 0; a < 6; a++) { if (Htz6) Htz6[a] = tr.Htz[a]; for (int b = 0; b < 6; b++) if (HtH36) HtH36[a * 6 + b] = tr.HtH[a * 7 + b]; }
  *error = tr.error; *n_meas = tr.n_meas;
  return 0;
}

// Full computeJacobianAndUpdateEKF (src/vio.cpp:784-802).  `seconds` = wall time around it (src/vio.cpp:1808-1812 window).
int orc_visual_update(const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels,
                      const double *inv_expo_list, int M, const StatePOD *state_in, const StatePOD *state_prop, StatePOD *state_out, float *errors,
                      void *trace /*VisualIterTrace[L*max_it]*/, int *n_trace, double *seconds, double *G361, double *RcwPcw12, const uint8_t *ref_imgs,
                      const int32_t *ref_img_idx, const double *ref_px, const double *ref_f, const double *ref_R, const double *ref_pos) {
  VIOManager vio; SubSparseMap sm; std::vector<VisualPoint> pts;
  setup_vio(vio, sm, pts, cfg, pos, warp_patch, search_levels, inv_expo_list, M, ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos);
  StatesGroup st, prop; st.from_pod(*state_in); prop.from_pod(*state_prop); vio.state = &st; vio.state_propagat = &prop;
  auto t0 = std::chrono::steady_clock::now();
  vio.computeJacobianAndUpdateEKF(img);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  st.to_pod(*state_out);
  if (errors) std::memcpy(errors, sm.errors.data(), M * 4);
  if (n_trace) *n_trace = (int)vio.trace_.size();
  if (trace) std::memcpy(trace, vio.trace_.data(), vio.trace_.size() * sizeof(VisualIterTrace));
  if (G361) std::memcpy(G361, vio.G.a, 361 * 8);
  if (RcwPcw12) { std::memcpy(RcwPcw12, vio.Rcw.a, 72); std::memcpy(RcwPcw12 + 9, vio.Pcw.a, 24); }
  return 0;
}

// calcBodyCov alone (src/voxel_map.cpp:15-34), for known-answer tests.
void orc_calc_body_cov(const double *pb3, float range_inc, float degree_inc, double deg2rad, double *cov9, double *pb_out3) {
  V3 pb = vec3(pb3[0], pb3[1], pb3[2]); M3 cov;
  calcBodyCov(pb, range_inc, degree_inc, cov, deg2rad);
  std::memcpy(cov9, cov.a, 72);
  if (pb_out3) std::memcpy(pb_out3, pb.a, 24);
}

// VoxelOctoTree::init_plane (src/voxel_map.cpp:55-135) on one group of points — checker of the device-side plane fit.
// out: the VoxelPlane fields init_plane writes.  Returns is_plane_.
struct PlaneFitPOD {
  double center[3], normal[3], y_normal[3], x_normal[3], covariance[9], plane_var[36];
  float radius, min_eigen_value, mid_eigen_value, max_eigen_value, d;
  int32_t points_size, is_plane, pad;
};
int orc_init_plane(const double *point_w, const double *var9, int n, float planer_threshold, PlaneFitPOD *out) {
  int id_counter = 0;
  VoxelOctoTree node(2, 0, 5, 50, planer_threshold, &id_counter);
  std::vector<MapPoint> pts((size_t)n);
  for (int i = 0; i < n; i++) {
    pts[i].point_w = vec3(point_w[3 * i], point_w[3 * i + 1], point_w[3 * i + 2]);
    std::memcpy(pts[i].var.a, var9 + 9 * (size_t)i, 72);
  }
  VoxelPlane pl;
  node.init_plane(pts, &pl);
  std::memset(out, 0, sizeof(*out));
  std::memcpy(out->center, pl.center_.a, 24); std::memcpy(out->normal, pl.normal_.a, 24);
  std::memcpy(out->y_normal, pl.y_normal_.a, 24); std::memcpy(out->x_normal, pl.x_normal_.a, 24);
  std::memcpy(out->covariance, pl.covariance_.a, 72); std::memcpy(out->plane_var, pl.plane_var_.a, 288);
  out->radius = pl.radius_; out->min_eigen_value = pl.min_eigen_value_; out->mid_eigen_value = pl.mid_eigen_value_; out->max_eigen_value = pl.max_eigen_value_;
  out->d = pl.d_; out->points_size = pl.points_size_; out->is_plane = pl.is_plane_ ? 1 : 0;
  return out->is_plane;
}

// the same over many groups (serial like UpdateVoxelMap, src/voxel_map.cpp:609-641); returns the seconds spent (bench.py cpu leg)
double orc_init_plane_batch(const double *point_w, const double *var9, const int *offsets, int n_groups, float planer_threshold, PlaneFitPOD *out) {
  const double t0 = omp_get_wtime();
  for (int g = 0; g < n_groups; g++)
    orc_init_plane(point_w + 3 * (size_t)offsets[g], var9 + 9 * (size_t)offsets[g], offsets[g + 1] - offsets[g], planer_threshold, out + g);
  return omp_get_wtime() - t0;
}

// Per-point tail of retrieveFromVisualSparseMap (src/vio.cpp:698-767) for n candidates (orc_warp.hpp).  Arrays are per candidate;
// ref_imgs holds n_ref images of cam.width x cam.height.  Outputs per candidate: accepted, search_level, error, ncc, A_cur_ref, and
// patch_wrap [n][L*64]; the survivors in candidate order are what the reference appends to visual_submap.  ref_id (may be NULL) =
// ref_ftr->id_: with !normal_en the warp of the first candidate carrying an id is reused by all later candidates with the same id
// (warp_map, src/vio.cpp:716-734).  Returns seconds spent.
struct orc_warp_cfg {
  double fx, fy, cx, cy; int32_t width, height;
  double R_cur[9], t_cur[3], inv_expo_cur;
  int32_t patch_pyrimid_level, normal_en, ncc_en, pad;
  double ncc_thre, outlier_threshold;
  double d[5]; int32_t distortion, pad2;          // vk::PinholeCamera radial-tangential coefficients (distortion = 0: pure pinhole)
};
double orc_warp_candidates(const orc_warp_cfg *c, const uint8_t *img, const uint8_t *ref_imgs, int n, const double *pos, const double *normal,
                           const int32_t *ref_img_idx, const double *ref_px, const double *ref_f, const double *ref_R, const double *ref_t,
                           const int32_t *ref_level, const double *ref_inv_expo, const int32_t *ref_id, int32_t *accepted, int32_t *search_level, float *error,
                           double *ncc, double *A4, float *patch_wrap) {
  WarpCfg cfg;
  cfg.cam.fx = c->fx; cfg.cam.fy = c->fy; cfg.cam.cx = c->cx; cfg.cam.cy = c->cy; cfg.cam.distortion = c->distortion; cfg.cam.width = c->width; cfg.cam.height = c->height;
  for (int k = 0; k < 5; k++) cfg.cam.d[k] = c->d[k];
  std::memcpy(cfg.R_cur.a, c->R_cur, 72); std::memcpy(cfg.t_cur.a, c->t_cur, 24);
  cfg.inv_expo_cur = c->inv_expo_cur; cfg.patch_pyrimid_level = c->patch_pyrimid_level; cfg.normal_en = c->normal_en; cfg.ncc_en = c->ncc_en;
  cfg.ncc_thre = c->ncc_thre; cfg.outlier_threshold = c->outlier_threshold;
  const size_t img_bytes = (size_t)c->width * c->height;
  const int L = c->patch_pyrimid_level;
  const double t0 = omp_get_wtime();
  std::unordered_map<int, std::pair<int, std::array<double, 4>>> warp_map;      // id_ -> (search_level, A_cur_ref); cleared per call (vio.cpp:369)
  for (int i = 0; i < n; i++) {
    WarpCand w;
    w.pos = vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); w.normal = vec3(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]);
    w.img_ref = ref_imgs + img_bytes * ref_img_idx[i];
    w.px_ref[0] = ref_px[2 * i]; w.px_ref[1] = ref_px[2 * i + 1];
    w.f_ref = vec3(ref_f[3 * i], ref_f[3 * i + 1], ref_f[3 * i + 2]);
    std::memcpy(w.R_ref.a, ref_R + 9 * (size_t)i, 72); w.t_ref = vec3(ref_t[3 * i], ref_t[3 * i + 1], ref_t[3 * i + 2]);
    w.level_ref = ref_level[i]; w.inv_expo_ref = ref_inv_expo[i];
    WarpOut o;
    if (!cfg.normal_en && ref_id) {
      auto iter_warp = warp_map.find(ref_id[i]);
      if (iter_warp != warp_map.end()) { o.search_level = iter_warp->second.first; std::memcpy(o.A, iter_warp->second.second.data(), 32); }
      else { warp_matrix(cfg, w, o); warp_map[ref_id[i]] = {o.search_level, {o.A[0], o.A[1], o.A[2], o.A[3]}}; }
    } else warp_matrix(cfg, w, o);
    warp_finish(cfg, img, w, patch_wrap + (size_t)i * L * 64, o);
    accepted[i] = o.accepted; search_level[i] = o.search_level; error[i] = o.error; ncc[i] = o.ncc;
    std::memcpy(A4 + 4 * (size_t)i, o.A, 32);
  }
  return omp_get_wtime() - t0;
}

// Per-scan pre-stage (orc_preprocess.hpp): undistortion in place, then the voxel-grid filter.  poses: n_poses x 22 doubles (Pose6D).
void orc_undistort(float *xyz, const float *curvature, int n, const double *poses22, int n_poses, const double *rot_end9, const double *pos_end3,
                   const double *extR9, const double *extT3) {
  std::vector<Pose6D> P((size_t)n_poses);
  for (int i = 0; i < n_poses; i++) std::memcpy(&P[i], poses22 + 22 * (size_t)i, sizeof(Pose6D));
  M3 Re, ER; V3 pe, Et;
  std::memcpy(Re.a, rot_end9, 72); std::memcpy(ER.a, extR9, 72); std::memcpy(pe.a, pos_end3, 24); std::memcpy(Et.a, extT3, 24);
  undistort_points(xyz, curvature, n, P.data(), n_poses, Re, pe, ER, Et);
}
static_assert(sizeof(Pose6D) == 22 * 8, "Pose6D is 22 doubles");
int orc_voxel_grid(const float *xyz, int n, float leaf, float *out_xyz /*capacity n*3*/) {
  std::vector<float> o;
  const int m = voxel_grid_filter(xyz, n, leaf, o);
  if (m > 0) std::memcpy(out_xyz, o.data(), o.size() * 4);
  return m;
}

// vk::PinholeCamera::cam2world / world2cam of the camera in a warp cfg (tests: round trip, second opinion)
void orc_cam2world(const orc_warp_cfg *c, double u, double v, double *xyz) {
  PinholeCam cam; cam.fx = c->fx; cam.fy = c->fy; cam.cx = c->cx; cam.cy = c->cy; cam.distortion = c->distortion; cam.width = c->width; cam.height = c->height;
  for (int k = 0; k < 5; k++) cam.d[k] = c->d[k];
  const V3 f = cam2world(cam, u, v);
  xyz[0] = f[0]; xyz[1] = f[1]; xyz[2] = f[2];
}
void orc_world2cam(const orc_warp_cfg *c, const double *xyz, double *px) {
  PinholeCam cam; cam.fx = c->fx; cam.fy = c->fy; cam.cx = c->cx; cam.cy = c->cy; cam.distortion = c->distortion; cam.width = c->width; cam.height = c->height;
  for (int k = 0; k < 5; k++) cam.d[k] = c->d[k];
  cam.world2cam(vec3(xyz[0], xyz[1], xyz[2]), px);
}

// Selection half of retrieveFromVisualSparseMap (orc_select.hpp).  keys: [n_pts][3] int64 feat_map keys (NULL: computed from pos with
// insertPointIntoVoxelMap's formula).  Returns seconds.
struct orc_select_cfg { double fx, fy, cx, cy; int32_t width, height; double R_cur[9], t_cur[3]; int32_t border, grid_size, grid_n_width, grid_n_height, patch_size_half, pad; double d[5]; int32_t distortion, raycast_en; };
// map (nullable): the LiDAR VoxelMap the RayCasting module looks into (plane_map of retrieveFromVisualSparseMap); add6 [add_cap][6]: center_, normal_ of
// visual_submap->add_from_voxel_map in push order, *n_add their number
double orc_visual_select_rc(const orc_select_cfg *c, const double *pg, int n_pg, const double *pos, const int64_t *keys, const uint8_t *active, int n_pts,
                            int32_t *cell_point, float *cell_dist, int32_t *cell_type, int32_t *discont, int32_t *in_fov, float *depth_img, void *map, double *add6, int add_cap,
                            int32_t *n_add);
double orc_visual_select(const orc_select_cfg *c, const double *pg, int n_pg, const double *pos, const int64_t *keys, const uint8_t *active, int n_pts,
                         int32_t *cell_point, float *cell_dist, int32_t *cell_type, int32_t *discont, int32_t *in_fov, float *depth_img) {
  return orc_visual_select_rc(c, pg, n_pg, pos, keys, active, n_pts, cell_point, cell_dist, cell_type, discont, in_fov, depth_img, nullptr, nullptr, 0, nullptr);
}
double orc_visual_select_rc(const orc_select_cfg *c, const double *pg, int n_pg, const double *pos, const int64_t *keys, const uint8_t *active, int n_pts,
                            int32_t *cell_point, float *cell_dist, int32_t *cell_type, int32_t *discont, int32_t *in_fov, float *depth_img, void *map, double *add6, int add_cap,
                            int32_t *n_add) {
  SelectCfg cfg;
  cfg.cam.fx = c->fx; cfg.cam.fy = c->fy; cfg.cam.cx = c->cx; cfg.cam.cy = c->cy; cfg.cam.distortion = c->distortion; cfg.cam.width = c->width; cfg.cam.height = c->height;
  for (int k = 0; k < 5; k++) cfg.cam.d[k] = c->d[k];
  std::memcpy(cfg.R_cur.a, c->R_cur, 72); std::memcpy(cfg.t_cur.a, c->t_cur, 24);
  cfg.border = c->border; cfg.grid_size = c->grid_size; cfg.grid_n_width = c->grid_n_width; cfg.grid_n_height = c->grid_n_height; cfg.patch_size_half = c->patch_size_half;
  std::vector<VisualMapPoint> pts((size_t)n_pts);
  for (int i = 0; i < n_pts; i++) {
    pts[i].pos = vec3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]); pts[i].active = active ? active[i] : 1;
    if (keys) for (int j = 0; j < 3; j++) pts[i].key[j] = keys[3 * (size_t)i + j]; else feat_map_key(pts[i].pos, pts[i].key);
  }
  cfg.raycast_en = c->raycast_en;
  MapHandle *mh = (MapHandle *)map;
  const PlaneLookup lookup = [mh](const int64_t key[3], const V3 &pw, V3 &center, V3 &normal) {
    auto it = mh->map.voxel_map_.find(VOXEL_LOCATION(key[0], key[1], key[2]));
    if (it == mh->map.voxel_map_.end()) return false;
    VoxelOctoTree *current_octo = it->second->find_correspond(pw);
    if (!current_octo->plane_ptr_->is_plane_) return false;
    center = current_octo->plane_ptr_->center_; normal = current_octo->plane_ptr_->normal_;
    return true;
  };
  std::vector<RayHit> add;
  const double t0 = omp_get_wtime();
  visual_select(cfg, pg, n_pg, pts.data(), n_pts, cell_point, cell_dist, cell_type, discont, in_fov, depth_img, mh ? &lookup : nullptr, &add);
  const double dt = omp_get_wtime() - t0;
  if (n_add) *n_add = (int32_t)add.size();
  for (int k = 0; k < (int)add.size() && k < add_cap && add6; k++) { std::memcpy(add6 + 6 * (size_t)k, add[k].center.a, 24); std::memcpy(add6 + 6 * (size_t)k + 3, add[k].normal.a, 24); }
  return dt;
}
void orc_feat_map_key(const double *pos3, int64_t *key3) { feat_map_key(vec3(pos3[0], pos3[1], pos3[2]), key3); }

// Reference-patch choice of retrieveFromVisualSparseMap (orc_choice.hpp) for the n_cells grid cells of one call.  cell_point / cell_discont: as
// produced by orc_visual_select.  The observations are a CSR table over the visual points: obs_offset[n_pts + 1]; per observation id_, T_f_w_
// (R row-major, t) and patch_ (64 floats); per point is_normal_initialized_ and ref_patch (GLOBAL observation index, -1 = !has_ref_patch_; updated in
// place like pt->ref_patch / has_ref_patch_).  cell_obs[c] = global index of ref_ftr, or -1 where the loop body `continue`s before the warp.
void orc_choose_ref(int normal_en, const double *R_cur9, const double *t_cur3, int n_cells, const int32_t *cell_point, const int32_t *cell_discont, const double *pos,
                    const int32_t *obs_offset, const int32_t *obs_id, const double *obs_R, const double *obs_t, const float *obs_patch,
                    const uint8_t *normal_initialized, int32_t *ref_patch, int32_t *cell_obs) {
  M3 Rc; V3 tc;
  std::memcpy(Rc.a, R_cur9, 72); std::memcpy(tc.a, t_cur3, 24);
  const V3 framepos = (Rc.T() * tc) * (-1.0);                       // new_frame_->pos()
  for (int c = 0; c < n_cells; c++) {
    cell_obs[c] = -1;
    const int p = cell_point[c];
    if (p < 0 || cell_discont[c]) continue;                         // vio.cpp:601-640
    if (!normal_initialized[p]) continue;                           // vio.cpp:650
    const int b = obs_offset[p], n = obs_offset[p + 1] - b;
    std::vector<ObsRef> obs((size_t)n);
    for (int k = 0; k < n; k++) {
      obs[k].id = obs_id[b + k]; std::memcpy(obs[k].R.a, obs_R + 9 * (size_t)(b + k), 72); std::memcpy(obs[k].t.a, obs_t + 3 * (size_t)(b + k), 24);
      obs[k].patch = obs_patch + 64 * (size_t)(b + k);
    }
    int chosen;
    if (normal_en) {
      int has = ref_patch[p] >= 0, ref = has ? ref_patch[p] - b : -1;
      chosen = choose_ref_by_patches(obs.data(), n, 64, has, ref);
      ref_patch[p] = has ? b + ref : -1;
    } else chosen = getCloseViewObs(framepos, vec3(pos[3 * p], pos[3 * p + 1], pos[3 * p + 2]), obs.data(), n);
    cell_obs[c] = chosen < 0 ? -1 : b + chosen;
  }
}

// IMU forward propagation (orc_imu.hpp).  steps: n x 8 doubles (gyr3, acc3, dt, offs_t); cfg: 13 doubles + 3 flags; poses out: n x 22 doubles.
struct orc_imu_cfg { double cov_gyr[3], cov_acc[3], cov_bias_gyr[3], cov_bias_acc[3], cov_inv_expo, G_m_s2, mean_acc_norm; int32_t ba_bg_est_en, gravity_est_en, exposure_estimate_en, first_call; };
double orc_imu_propagate(const StatePOD *in, const double *steps8, int n, const orc_imu_cfg *c, StatePOD *out, double *poses22) {
  StatesGroup st; st.from_pod(*in);
  ImuCfg cfg;
  std::memcpy(cfg.cov_gyr, c->cov_gyr, 24); std::memcpy(cfg.cov_acc, c->cov_acc, 24); std::memcpy(cfg.cov_bias_gyr, c->cov_bias_gyr, 24); std::memcpy(cfg.cov_bias_acc, c->cov_bias_acc, 24);
  cfg.cov_inv_expo = c->cov_inv_expo; cfg.G_m_s2 = c->G_m_s2; cfg.mean_acc_norm = c->mean_acc_norm;
  cfg.ba_bg_est_en = c->ba_bg_est_en; cfg.gravity_est_en = c->gravity_est_en; cfg.exposure_estimate_en = c->exposure_estimate_en; cfg.first_call = c->first_call;
  std::vector<ImuStep> S((size_t)n); std::vector<Pose6D> P((size_t)n);
  static_assert(sizeof(ImuStep) == 64, "ImuStep is 8 doubles");
  if (n) std::memcpy(S.data(), steps8, (size_t)n * 64);
  const double t0 = omp_get_wtime();
  imu_propagate(st, S.data(), n, cfg, P.data());
  const double dt = omp_get_wtime() - t0;
  st.to_pod(*out);
  if (n) std::memcpy(poses22, P.data(), (size_t)n * sizeof(Pose6D));
  return dt;
}

// State algebra (common_lib.h:182-206) and the 19x19 inverse, for known-answer tests.
void orc_state_boxplus(const StatePOD *s, const double *d19, StatePOD *out) { StatesGroup g; g.from_pod(*s); VState d; std::memcpy(d.a, d19, 152); g += d; g.to_pod(*out); }
void orc_state_boxminus(const StatePOD *a, const StatePOD *b, double *out19) { StatesGroup ga, gb; ga.from_pod(*a); gb.from_pod(*b); VState d = ga - gb; std::memcpy(out19, d.a, 152); }
int orc_inverse19(const double *A361, double *inv361) { MState A, I; std::memcpy(A.a, A361, 361 * 8); bool ok = inverse_lu<19>(A, I); std::memcpy(inv361, I.a, 361 * 8); return ok ? 0 : -1; }
void orc_so3_exp(const double *v3, double *R9) { M3 R = Exp(v3[0], v3[1], v3[2]); std::memcpy(R9, R.a, 72); }
void orc_so3_log(const double *R9, double *v3) { M3 R; std::memcpy(R.a, R9, 72); V3 v = Log(R); std::memcpy(v3, v.a, 24); }

} // extern "C"
