// TEST INFRASTRUCTURE ONLY — the pin of the oracle.  Never linked into, imported by, or executed from the product path.
//
// extern "C" driver around the REFERENCE'S OWN translation units (/root/reference/src/{voxel_map,vio,frame,visual_point}.cpp, compiled
// textually unmodified by oracle/ref_build/Makefile against the stand-in headers in stubs/).  The entry points carry the same names and
// signatures as the corresponding ones of oracle/orc_api.cpp, so oracle/orc.py drives either library (`lib=` argument) and
// tests/test_ref_pin_cpu.py can feed one scenario to both and compare.  This file only moves arrays into and out of the reference's
// classes (VoxelMapManager, VoxelOctoTree, VIOManager, Frame, Feature, VisualPoint); every arithmetic statement that runs is the
// reference's.  Quantities that are locals of the reference's functions (per-iteration H rows, R_inv, z, H_sub) cannot be read out of
// unmodified code; the tests reach them through truncated runs (max_iterations = 1, 2, ...) instead.
#include "vio.h"
#include <vikit/equidistant_camera.h>
#include <chrono>
#include <cstring>
#include <sstream>
#include <fcntl.h>
#include <unistd.h>

namespace {

struct StatePOD {            // oracle/orc_state.hpp
  double rot[9], pos[3], inv_expo, vel[3], bg[3], ba[3], grav[3], cov[361];
};
struct orc_lidar_cfg {       // oracle/orc_api.cpp
  int max_iterations, max_layer;
  double sigma_num, dept_err, beam_err, voxel_size, deg2rad;
  double extR[9], extT[3];
  int num_threads, pad;
};
struct orc_visual_cfg {
  double fx, fy, cx, cy, d[5];
  int distortion, width, height, patch_pyrimid_level, max_iterations, exposure_estimate_en, inverse_composition_en, num_threads;
  double img_point_cov;
  double Rcl[9], Pcl[3], extR[9], extT[3];
};
struct LidarIterTrace { int n_eff; double total_residual; double HtH[36], Htz[6]; double solution[19]; int converged, stopped; };
struct VisualIterTrace { int level, iteration, accepted, n_meas; float error; double HtH[49], Htz[7], solution[19]; };
struct PlaneFitPOD {
  double center[3], normal[3], y_normal[3], x_normal[3], covariance[9], plane_var[36];
  float radius, min_eigen_value, mid_eigen_value, max_eigen_value, d;
  int32_t points_size, is_plane, pad;
};

template <class M> void rm_in(M &m, const double *a) { for (int i = 0; i < (int)m.rows(); i++) for (int j = 0; j < (int)m.cols(); j++) m(i, j) = a[i * m.cols() + j]; }
template <class M> void rm_out(const M &m, double *a) { for (int i = 0; i < (int)m.rows(); i++) for (int j = 0; j < (int)m.cols(); j++) a[i * m.cols() + j] = m(i, j); }

void from_pod(StatesGroup &s, const StatePOD &p) {
  rm_in(s.rot_end, p.rot); rm_in(s.pos_end, p.pos); s.inv_expo_time = p.inv_expo; rm_in(s.vel_end, p.vel); rm_in(s.bias_g, p.bg); rm_in(s.bias_a, p.ba);
  rm_in(s.gravity, p.grav); rm_in(s.cov, p.cov);
}
void to_pod(const StatesGroup &s, StatePOD &p) {
  rm_out(s.rot_end, p.rot); rm_out(s.pos_end, p.pos); p.inv_expo = s.inv_expo_time; rm_out(s.vel_end, p.vel); rm_out(s.bias_g, p.bg); rm_out(s.bias_a, p.ba);
  rm_out(s.gravity, p.grav); rm_out(s.cov, p.cov);
}

struct MapHandle {
  VoxelMapConfig cfg;
  std::unordered_map<VOXEL_LOCATION, VoxelOctoTree *> voxel_map;
  V3D last_slide_position = V3D(0, 0, 0);                   // VoxelMapManager::last_slide_position (voxel_map.h:209) lives across calls
  std::unordered_map<std::string, int> plane_by_center;     // bytes of center_ -> plane index of the flat map (PointToPlane carries no plane pointer)
  ~MapHandle() { for (auto &kv : voxel_map) delete kv.second; }
};
std::string center_key(const Eigen::Vector3d &c) { double v[3] = {c[0], c[1], c[2]}; return std::string((const char *)v, 24); }

void count_nodes(const VoxelOctoTree *t, int &n_nodes, int &n_planes) {
  n_nodes++;
  if (t->plane_ptr_->is_plane_) n_planes++;
  for (int i = 0; i < 8; i++) if (t->leaves_[i]) count_nodes(t->leaves_[i], n_nodes, n_planes);
}
struct Exporter {
  int32_t *node_plane, *node_child; double *pn, *pc, *pv; float *pd, *pr; int n_nodes = 0, n_planes = 0;
  int visit(const VoxelOctoTree *t) {
    int me = n_nodes++;
    if (t->plane_ptr_->is_plane_) {
      const VoxelPlane *p = t->plane_ptr_;
      int pi = n_planes++;
      for (int k = 0; k < 3; k++) { pn[pi * 3 + k] = p->normal_[k]; pc[pi * 3 + k] = p->center_[k]; }
      rm_out(p->plane_var_, pv + (size_t)pi * 36);
      pd[pi] = p->d_; pr[pi] = p->radius_;
      node_plane[me] = pi;
    } else node_plane[me] = -1;
    for (int i = 0; i < 8; i++) node_child[(size_t)me * 8 + i] = -1;
    for (int i = 0; i < 8; i++) if (t->leaves_[i]) { int c = visit(t->leaves_[i]); node_child[(size_t)me * 8 + i] = c; }
    return me;
  }
};

// std::cout of the reference's own diagnostics is captured, not printed; its printf output goes to /dev/null for the duration
struct CoutCapture {
  std::ostringstream buf; std::streambuf *old; int saved_fd = -1;
  CoutCapture() : old(std::cout.rdbuf(buf.rdbuf())) {
    std::fflush(stdout);
    saved_fd = dup(1);
    int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) { dup2(nul, 1); close(nul); }
  }
  ~CoutCapture() { std::cout.rdbuf(old); std::fflush(stdout); if (saved_fd >= 0) { dup2(saved_fd, 1); close(saved_fd); } }
};

} // namespace

// vikit symbols that the compiled sources name outside the parity path (declared in stubs/vikit/vision.h)
namespace vk {
float shiTomasiScore(const cv::Mat &, int, int) { return 0.f; }
void halfSample(const cv::Mat &, cv::Mat &) {}
} // namespace vk

extern "C" {

const char *ref_describe() {
  return "reference translation units voxel_map.cpp vio.cpp frame.cpp visual_point.cpp compiled unmodified; Eigen/PCL/OpenCV/Sophus/vikit/ROS replaced by oracle/ref_build/stubs; "
#ifdef MP_EN
         "MP_EN";
#else
         "serial (MP_EN undefined)";
#endif
}
int ref_mp_proc_num() {
#ifdef MP_EN
  return MP_PROC_NUM;
#else
  return 1;
#endif
}

void orc_set_sum_model(int, int, int) {}
int orc_sizeof_lidar_trace() { return (int)sizeof(LidarIterTrace); }
int orc_sizeof_visual_trace() { return (int)sizeof(VisualIterTrace); }
int orc_sizeof_state() { return (int)sizeof(StatePOD); }

void *orc_map_create(double voxel_size, int max_layer, const int *layer_init_num5, int max_points_num, double planer_threshold) {
  MapHandle *h = new MapHandle;
  h->cfg = VoxelMapConfig();
  h->cfg.max_voxel_size_ = voxel_size; h->cfg.max_layer_ = max_layer; h->cfg.layer_init_num_.assign(layer_init_num5, layer_init_num5 + 5);
  h->cfg.max_points_num_ = max_points_num; h->cfg.planner_threshold_ = planer_threshold;
  h->cfg.max_iterations_ = 5; h->cfg.sigma_num_ = 3; h->cfg.beam_err_ = 0.05; h->cfg.dept_err_ = 0.02; h->cfg.is_pub_plane_map_ = false;
  h->cfg.sliding_thresh = 8; h->cfg.map_sliding_en = false; h->cfg.half_map_size = 100;
  return h;
}
void orc_map_destroy(void *m) { delete (MapHandle *)m; }

static void to_points(const double *pw, const double *var9, int n, std::vector<pointWithVar> &pts) {
  pts.resize(n);
  for (int i = 0; i < n; i++) { for (int k = 0; k < 3; k++) pts[i].point_w[k] = pw[(size_t)i * 3 + k]; rm_in(pts[i].var, var9 + (size_t)i * 9); }
}

// VoxelMapManager::UpdateVoxelMap (voxel_map.cpp:609-641) on caller-supplied (point_w, var)
void orc_map_update(void *m, const double *pw, const double *var9, int n) {
  MapHandle *h = (MapHandle *)m;
  std::vector<pointWithVar> pts; to_points(pw, var9, n, pts);
  VoxelMapManager mgr(h->cfg, h->voxel_map);
  mgr.UpdateVoxelMap(pts);
  h->voxel_map = mgr.voxel_map_;
}

// VoxelMapManager::BuildVoxelMap (voxel_map.cpp:532-591).  It forms point_w / var itself from feats_down_world_ / feats_down_body_ /
// state_ / extR_, so the inputs are those (float32 clouds), not (point_w, var) as in oracle/orc_api.cpp:orc_map_build.
void ref_map_build_from_scan(void *m, const float *body_xyz, const float *world_xyz, int n, const StatePOD *state, const double *extR9, double dept_err, double beam_err) {
  MapHandle *h = (MapHandle *)m;
  h->cfg.dept_err_ = dept_err; h->cfg.beam_err_ = beam_err;
  VoxelMapManager mgr(h->cfg, h->voxel_map);
  rm_in(mgr.extR_, extR9); mgr.extT_ = V3D::Zero();
  from_pod(mgr.state_, *state);
  for (int i = 0; i < n; i++) {
    PointType b; b.x = body_xyz[3 * i]; b.y = body_xyz[3 * i + 1]; b.z = body_xyz[3 * i + 2]; mgr.feats_down_body_->points.push_back(b);
    PointType w; w.x = world_xyz[3 * i]; w.y = world_xyz[3 * i + 1]; w.z = world_xyz[3 * i + 2]; mgr.feats_down_world_->points.push_back(w);
  }
  mgr.feats_down_size_ = n;
  mgr.BuildVoxelMap();
  h->voxel_map = mgr.voxel_map_;
}

// VoxelMapManager::mapSliding / clearMemOutOfMap (voxel_map.cpp:924-972); returns the number of root voxels removed, -1 below the threshold
int orc_map_slide(void *m, const double *position_last, double sliding_thresh, int half_map_size) {
  MapHandle *h = (MapHandle *)m;
  h->cfg.sliding_thresh = sliding_thresh; h->cfg.half_map_size = half_map_size; h->cfg.map_sliding_en = true;
  VoxelMapManager mgr(h->cfg, h->voxel_map);
  for (int k = 0; k < 3; k++) mgr.position_last_[k] = position_last[k];
  mgr.last_slide_position = h->last_slide_position;
  const size_t before = mgr.voxel_map_.size();
  const bool below = (mgr.position_last_ - mgr.last_slide_position).norm() < sliding_thresh;      // the early return at voxel_map.cpp:926-930 reports nothing
  { CoutCapture cap; mgr.mapSliding(); }
  h->voxel_map = mgr.voxel_map_; h->last_slide_position = mgr.last_slide_position;
  return below ? -1 : (int)(before - h->voxel_map.size());
}

void orc_map_counts(void *m, int *n_roots, int *n_nodes, int *n_planes) {
  MapHandle *h = (MapHandle *)m; int nn = 0, np = 0;
  for (auto &kv : h->voxel_map) count_nodes(kv.second, nn, np);
  *n_roots = (int)h->voxel_map.size(); *n_nodes = nn; *n_planes = np;
}

void orc_map_export(void *m, int64_t *keys, int32_t *root_node, double *root_center, float *root_quarter, int32_t *node_plane, int32_t *node_child,
                    double *plane_normal, double *plane_center, double *plane_var, float *plane_d, float *plane_radius) {
  MapHandle *h = (MapHandle *)m;
  Exporter ex{node_plane, node_child, plane_normal, plane_center, plane_var, plane_d, plane_radius};
  int r = 0;
  for (auto &kv : h->voxel_map) {
    keys[r * 3 + 0] = kv.first.x; keys[r * 3 + 1] = kv.first.y; keys[r * 3 + 2] = kv.first.z;
    for (int k = 0; k < 3; k++) root_center[r * 3 + k] = kv.second->voxel_center_[k];
    root_quarter[r] = kv.second->quater_length_;
    root_node[r] = ex.visit(kv.second);
    r++;
  }
}

static VoxelOctoTree *build_from_flat(MapHandle *h, int node, int layer, const int32_t *node_plane, const int32_t *node_child, const double *pn, const double *pc,
                                      const double *pv, const float *pd, const float *pr) {
  VoxelOctoTree *t = new VoxelOctoTree(h->cfg.max_layer_, layer, 5, h->cfg.max_points_num_, (float)h->cfg.planner_threshold_);
  t->layer_init_num_ = h->cfg.layer_init_num_;
  t->init_octo_ = true;
  int pi = node_plane[node];
  if (pi >= 0) {
    VoxelPlane *p = t->plane_ptr_;
    for (int k = 0; k < 3; k++) { p->normal_[k] = pn[pi * 3 + k]; p->center_[k] = pc[pi * 3 + k]; }
    rm_in(p->plane_var_, pv + (size_t)pi * 36);
    p->d_ = pd[pi]; p->radius_ = pr[pi]; p->is_plane_ = true; p->is_init_ = true;
    h->plane_by_center.emplace(center_key(p->center_), pi);      // two voxels holding byte-identical plane records cannot be told apart from a PointToPlane: lowest index reported
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
    h->voxel_map[VOXEL_LOCATION(keys[r * 3], keys[r * 3 + 1], keys[r * 3 + 2])] = t;
  }
  return h;
}

// VoxelMapManager::StateEstimation (voxel_map.cpp:338-511) exactly as LIVMapper::handleLIO drives it (LIVMapper.cpp:357-372): the members it
// reads are set, the method is called, the members it writes are read back.  n_iters / n_eff come from the reference's own "[ LIO ]" line.
int orc_lidar_state_estimation(void *m, const orc_lidar_cfg *cfg, const float *xyz, int n, const StatePOD *state_in, const StatePOD *state_prop,
                               StatePOD *state_out, void *trace, int *n_iters, double *seconds, int32_t *match_plane, float *dis, float *pw, double *var,
                               double *body_cov, double *cross_mat, double *normal, double *Rinv, double *Hrow) {
  MapHandle *h = (MapHandle *)m;
  if (std::fabs(cfg->deg2rad - DEG2RAD(1.0)) > 0) return -2;        // the macro is compiled in (stubs/pcl/point_types.h)
  VoxelMapConfig c = h->cfg;
  c.max_iterations_ = cfg->max_iterations; c.max_layer_ = cfg->max_layer; c.sigma_num_ = cfg->sigma_num; c.dept_err_ = cfg->dept_err; c.beam_err_ = cfg->beam_err;
  c.max_voxel_size_ = cfg->voxel_size;
  VoxelMapManager mgr(c, h->voxel_map);
  rm_in(mgr.extR_, cfg->extR); rm_in(mgr.extT_, cfg->extT);
  for (int i = 0; i < n; i++) { PointType p; p.x = xyz[(size_t)i * 3]; p.y = xyz[(size_t)i * 3 + 1]; p.z = xyz[(size_t)i * 3 + 2]; mgr.feats_down_body_->points.push_back(p); }
  mgr.feats_down_size_ = n;
  from_pod(mgr.state_, *state_in);
  StatesGroup prop; from_pod(prop, *state_prop);
  std::string log;
  {
    CoutCapture cap;
    auto t0 = std::chrono::steady_clock::now();
    mgr.StateEstimation(prop);
    auto t1 = std::chrono::steady_clock::now();
    if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
    log = cap.buf.str();
  }
  to_pod(mgr.state_, *state_out);
  int it = 0;
  LidarIterTrace *tr = (LidarIterTrace *)trace;
  for (size_t pos = 0; (pos = log.find("[ LIO ]", pos)) != std::string::npos; pos++) {
    if (tr) {
      std::memset(&tr[it], 0, sizeof(LidarIterTrace));
      size_t a = log.find("effective feature num: ", pos), b = log.find("average residual: ", pos);
      if (a != std::string::npos) tr[it].n_eff = std::atoi(log.c_str() + a + 23);
      if (b != std::string::npos) tr[it].total_residual = std::atof(log.c_str() + b + 18) * tr[it].n_eff;    // printed with 6 significant digits only
    }
    it++;
  }
  if (n_iters) *n_iters = it;
  if (match_plane) for (int i = 0; i < n; i++) match_plane[i] = -1;
  if (dis) for (int i = 0; i < n; i++) dis[i] = 0;
  if (Rinv) for (int i = 0; i < n; i++) Rinv[i] = 0;               // locals of StateEstimation: not observable
  if (Hrow) for (size_t i = 0; i < (size_t)n * 6; i++) Hrow[i] = 0;
  for (int i = 0; i < n; i++) {
    const pointWithVar &pv = mgr.pv_list_[i];
    if (pw) for (int k = 0; k < 3; k++) pw[(size_t)i * 3 + k] = (float)pv.point_w[k];
    if (var) rm_out(pv.var, var + (size_t)i * 9);
    if (body_cov) rm_out(mgr.body_cov_list_[i], body_cov + (size_t)i * 9);
    if (cross_mat) rm_out(mgr.cross_mat_list_[i], cross_mat + (size_t)i * 9);
    if (normal) for (int k = 0; k < 3; k++) normal[(size_t)i * 3 + k] = pv.normal[k];
  }
  int i = 0;
  for (size_t j = 0; j < mgr.ptpl_list_.size(); j++) {             // ptpl_list_ keeps the input order (voxel_map.cpp:707-710)
    const PointToPlane &pl = mgr.ptpl_list_[j];
    while (i < n && !(mgr.pv_list_[i].point_b == pl.point_b_)) i++;
    if (i >= n) return -3;
    if (match_plane) { auto f = h->plane_by_center.find(center_key(pl.center_)); match_plane[i] = (f == h->plane_by_center.end()) ? -2 : f->second; }
    if (dis) dis[i] = pl.dis_to_plane_;
    i++;
  }
  if ((int)mgr.ptpl_list_.size() != mgr.effct_feat_num_) return -4;
  return 0;
}

namespace {
struct VioRig {
  VIOManager vio;
  std::unique_ptr<vk::AbstractCamera> cam;
  std::vector<std::unique_ptr<VisualPoint>> pts;
  StatesGroup st, prop;
  cv::Mat img;
  std::vector<cv::Mat> ref_imgs;
  ~VioRig() { vio.visual_submap->voxel_points.clear(); }
};

void setup_vio(VioRig &r, const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels,
               const double *inv_expo_list, int M, const uint8_t *ref_imgs, const int32_t *ref_img_idx, const double *ref_px, const double *ref_f, const double *ref_R,
               const double *ref_pos) {
  VIOManager &vio = r.vio;
  if (cfg->distortion == 2) r.cam.reset(new vk::EquidistantCamera(cfg->width, cfg->height, 1.0, cfg->fx, cfg->fy, cfg->cx, cfg->cy, cfg->d[0], cfg->d[1], cfg->d[2], cfg->d[3]));
  else r.cam.reset(new vk::PinholeCamera(cfg->width, cfg->height, 1.0, cfg->fx, cfg->fy, cfg->cx, cfg->cy, cfg->distortion ? cfg->d[0] : 0.0, cfg->distortion ? cfg->d[1] : 0.0,
                                         cfg->distortion ? cfg->d[2] : 0.0, cfg->distortion ? cfg->d[3] : 0.0, cfg->distortion ? cfg->d[4] : 0.0));
  vio.cam = r.cam.get();
  vio.grid_size = 5; vio.grid_n_height = 17; vio.patch_size = 8; vio.patch_pyrimid_level = cfg->patch_pyrimid_level; vio.max_iterations = cfg->max_iterations;
  vio.img_point_cov = cfg->img_point_cov; vio.exposure_estimate_en = cfg->exposure_estimate_en; vio.inverse_composition_en = cfg->inverse_composition_en;
  vio.normal_en = true; vio.raycast_en = false; vio.ncc_en = false; vio.colmap_output_en = false; vio.has_ref_patch_cache = false; vio.plot_flag = false;
  vio.outlier_threshold = 1000; vio.ncc_thre = 0;
  M3D extR, Rcl; V3D extT; rm_in(extR, cfg->extR); rm_in(extT, cfg->extT);
  std::vector<double> Rv(cfg->Rcl, cfg->Rcl + 9), Pv(cfg->Pcl, cfg->Pcl + 3);
  vio.setImuToLidarExtrinsic(extT, extR);        // LIVMapper.cpp:133
  vio.setLidarToCameraExtrinsic(Rv, Pv);         // LIVMapper.cpp:134
  vio.state = &r.st; vio.state_propagat = &r.prop;
  vio.initializeVIO();                           // vio.cpp:41-159 (prints the intrinsics)
  r.img = cv::Mat(cfg->height, cfg->width, CV_8UC1); std::memcpy(r.img.data, img, (size_t)cfg->width * cfg->height);
  vio.new_frame_.reset(new Frame(vio.cam, r.img));
  const int L = cfg->patch_pyrimid_level;
  SubSparseMap &sm = *vio.visual_submap;
  sm.errors.assign(M, 0.f); sm.propa_errors.assign(M, 0.f); sm.warp_patch.resize(M); sm.search_levels.resize(M); sm.voxel_points.resize(M); sm.inv_expo_list.resize(M);
  r.pts.resize(M);
  if (ref_imgs && ref_img_idx) {
    int n_ref = 0; for (int i = 0; i < M; i++) n_ref = std::max(n_ref, ref_img_idx[i] + 1);
    r.ref_imgs.resize(n_ref);
    for (int k = 0; k < n_ref; k++) { r.ref_imgs[k] = cv::Mat(cfg->height, cfg->width, CV_8UC1); std::memcpy(r.ref_imgs[k].data, ref_imgs + (size_t)k * cfg->width * cfg->height, (size_t)cfg->width * cfg->height); }
  }
  for (int i = 0; i < M; i++) {
    r.pts[i].reset(new VisualPoint(V3D(pos[(size_t)i * 3], pos[(size_t)i * 3 + 1], pos[(size_t)i * 3 + 2])));
    sm.voxel_points[i] = r.pts[i].get();
    sm.warp_patch[i].assign(warp_patch + (size_t)i * L * 64, warp_patch + (size_t)(i + 1) * L * 64);
    sm.search_levels[i] = search_levels[i];
    sm.inv_expo_list[i] = inv_expo_list[i];
    if (ref_imgs && ref_img_idx) {
      // Feature::pos() = T_f_w_.inverse().translation() = -(R^T t) must return ref_pos: t = -(R^T)^-1 ref_pos (the yaml extrinsics are
      // orthonormal to 1e-6 only, so R^T is inverted rather than transposed; pos() then agrees with ref_pos to rounding)
      M3D Rrw; rm_in(Rrw, ref_R + (size_t)i * 9); V3D c(ref_pos[i * 3], ref_pos[i * 3 + 1], ref_pos[i * 3 + 2]);
      Feature *f = new Feature(r.pts[i].get(), new float[64](), V2D(ref_px[i * 2], ref_px[i * 2 + 1]), V3D(ref_f[i * 3], ref_f[i * 3 + 1], ref_f[i * 3 + 2]), SE3(Rrw, -(Rrw.transpose().inverse() * c)), 0);
      f->img_ = r.ref_imgs[ref_img_idx[i]]; f->id_ = i; f->inv_expo_time_ = 1.0;
      r.pts[i]->addFrameRef(f); r.pts[i]->ref_patch = f; r.pts[i]->has_ref_patch_ = true;
    }
  }
  vio.total_points = M;
}
} // namespace

// VIOManager::computeJacobianAndUpdateEKF (vio.cpp:784-802) as VIOManager::processFrame drives it (vio.cpp:1808-1812)
int orc_visual_update(const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels,
                      const double *inv_expo_list, int M, const StatePOD *state_in, const StatePOD *state_prop, StatePOD *state_out, float *errors, void *trace,
                      int *n_trace, double *seconds, double *G361, double *RcwPcw12, const uint8_t *ref_imgs, const int32_t *ref_img_idx, const double *ref_px,
                      const double *ref_f, const double *ref_R, const double *ref_pos) {
  (void)trace;
  VioRig r;
  from_pod(r.st, *state_in); from_pod(r.prop, *state_prop);
  { CoutCapture cap; std::fflush(stdout); setup_vio(r, cfg, img, pos, warp_patch, search_levels, inv_expo_list, M, ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos); }
  r.vio.G.setZero(); r.vio.H_T_H.setZero();
  CoutCapture cap;
  auto t0 = std::chrono::steady_clock::now();
  r.vio.computeJacobianAndUpdateEKF(r.img);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  to_pod(r.st, *state_out);
  if (errors) std::memcpy(errors, r.vio.visual_submap->errors.data(), (size_t)M * 4);
  if (n_trace) *n_trace = 0;                      // per-step quantities are locals of updateState: not observable
  if (G361) rm_out(r.vio.G, G361);
  if (RcwPcw12) { rm_out(r.vio.Rcw, RcwPcw12); rm_out(r.vio.Pcw, RcwPcw12 + 9); }
  return 0;
}

// one level: VIOManager::updateState / updateStateInverse (vio.cpp:1520-1688 / 1398-1518) called directly (public members, vio.h:143-144)
int ref_visual_level(const orc_visual_cfg *cfg, const uint8_t *img, const double *pos, const float *warp_patch, const int32_t *search_levels, const double *inv_expo_list,
                     int M, int level, const StatePOD *state_in, const StatePOD *state_prop, StatePOD *state_out, float *errors, double *G361, double *HtH361,
                     const uint8_t *ref_imgs, const int32_t *ref_img_idx, const double *ref_px, const double *ref_f, const double *ref_R, const double *ref_pos) {
  VioRig r;
  from_pod(r.st, *state_in); from_pod(r.prop, *state_prop);
  { CoutCapture cap; setup_vio(r, cfg, img, pos, warp_patch, search_levels, inv_expo_list, M, ref_imgs, ref_img_idx, ref_px, ref_f, ref_R, ref_pos); }
  r.vio.G.setZero(); r.vio.H_T_H.setZero();
  CoutCapture cap;
  if (cfg->inverse_composition_en) { r.vio.has_ref_patch_cache = false; r.vio.updateStateInverse(r.img, level); }
  else r.vio.updateState(r.img, level);
  to_pod(r.st, *state_out);
  if (errors) std::memcpy(errors, r.vio.visual_submap->errors.data(), (size_t)M * 4);
  if (G361) rm_out(r.vio.G, G361);
  if (HtH361) rm_out(r.vio.H_T_H, HtH361);
  return 0;
}

// VIOManager::retrieveFromVisualSparseMap (vio.cpp:352-782) on a visual map given as flat arrays (scenarios.synth.RetrieveChainScenario): feat_map of VisualPoints with
// their Feature lists (obs_ in the given order), the current Frame, resetGrid, then the call as processFrame makes it (vio.cpp:1802-1808).  raycast_en = false.
struct ref_retrieve_cfg {
  double fx, fy, cx, cy, d[5]; int32_t distortion, width, height;
  double R_cur[9], t_cur[3], inv_expo_cur;
  int32_t patch_pyrimid_level, normal_en, ncc_en, border, grid_size, grid_n_height;
  double ncc_thre, outlier_threshold;
};
// the RayCasting module (vio.cpp:80-118, 487-591) of the NEXT ref_visual_retrieve call: raycast_en, the LiDAR VoxelMap handed to it as plane_map (a MapHandle of this
// library, nullable) and where visual_submap->add_from_voxel_map (center_, normal_ per entry) is copied to
namespace { struct RaycastSetup { int en = 0; void *map = nullptr; double *add6 = nullptr; int cap = 0; int32_t *n_add = nullptr; } g_rc; }
void ref_visual_retrieve_raycast(int raycast_en, void *map, double *add6, int add_cap, int32_t *n_add) { g_rc.en = raycast_en; g_rc.map = map; g_rc.add6 = add6; g_rc.cap = add_cap; g_rc.n_add = n_add; }
int ref_visual_retrieve(const ref_retrieve_cfg *c, const uint8_t *img, const uint8_t *ref_imgs, int n_ref, const double *pg, int n_pg, int n_pts, const double *pos,
                        const double *normal, const int64_t *keys, const uint8_t *active, const uint8_t *ninit, const int32_t *ref_patch_in, const int32_t *obs_offset,
                        const int32_t *obs_id, const int32_t *obs_img_idx, const int32_t *obs_level, const double *obs_px, const double *obs_f, const double *obs_R,
                        const double *obs_t, const double *obs_inv_expo, const float *obs_patch,
                        int32_t *cell_type, int32_t *cell_point, float *cell_dist, int32_t *ref_patch_out, int32_t *n_sub, int32_t *sub_point, int32_t *sub_obs,
                        int32_t *sub_search, float *sub_error, float *sub_patch /*[n_sub][L][64]*/, double *sub_inv_expo) {
  VIOManager vio;
  std::unique_ptr<vk::AbstractCamera> cam;
  if (c->distortion == 2) cam.reset(new vk::EquidistantCamera(c->width, c->height, 1.0, c->fx, c->fy, c->cx, c->cy, c->d[0], c->d[1], c->d[2], c->d[3]));
  else cam.reset(new vk::PinholeCamera(c->width, c->height, 1.0, c->fx, c->fy, c->cx, c->cy, c->distortion ? c->d[0] : 0.0, c->distortion ? c->d[1] : 0.0, c->distortion ? c->d[2] : 0.0,
                                       c->distortion ? c->d[3] : 0.0, c->distortion ? c->d[4] : 0.0));
  vio.cam = cam.get();
  StatesGroup st, prop; st.inv_expo_time = c->inv_expo_cur;
  vio.state = &st; vio.state_propagat = &prop;
  vio.grid_size = c->grid_size; vio.grid_n_height = c->grid_n_height; vio.patch_size = 8; vio.patch_pyrimid_level = c->patch_pyrimid_level; vio.max_iterations = 5;
  vio.img_point_cov = 100; vio.exposure_estimate_en = true; vio.inverse_composition_en = false; vio.normal_en = c->normal_en != 0; vio.raycast_en = g_rc.en != 0; vio.ncc_en = c->ncc_en != 0;
  vio.colmap_output_en = false; vio.has_ref_patch_cache = false; vio.plot_flag = false; vio.outlier_threshold = c->outlier_threshold; vio.ncc_thre = c->ncc_thre;
  vio.Rcl = M3D::Identity(); vio.Rli = M3D::Identity(); vio.Pcl = V3D::Zero(); vio.Pli = V3D::Zero();
  { CoutCapture cap; vio.initializeVIO(); }
  vio.border = c->border;
  const size_t bytes = (size_t)c->width * c->height;
  cv::Mat cur(c->height, c->width, CV_8UC1); std::memcpy(cur.data, img, bytes);
  std::vector<cv::Mat> refs(n_ref);
  for (int k = 0; k < n_ref; k++) { refs[k] = cv::Mat(c->height, c->width, CV_8UC1); std::memcpy(refs[k].data, ref_imgs + bytes * k, bytes); }
  vio.new_frame_.reset(new Frame(vio.cam, cur));
  { M3D R; V3D t; rm_in(R, c->R_cur); rm_in(t, c->t_cur); vio.new_frame_->T_f_w_ = SE3(R, t); }
  std::vector<VisualPoint *> pts(n_pts, nullptr);
  std::unordered_map<const Feature *, int> obs_index;
  std::unordered_map<const VisualPoint *, int> pt_index;
  for (int i = 0; i < n_pts; i++) {
    VisualPoint *pt = new VisualPoint(V3D(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
    pt->normal_ = V3D(normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]); pt->previous_normal_ = pt->normal_;
    pt->is_normal_initialized_ = ninit[i] != 0;
    pt->ref_patch = nullptr;
    if (active[i]) {
      for (int k = obs_offset[i]; k < obs_offset[i + 1]; k++) {
        M3D R; rm_in(R, obs_R + (size_t)k * 9);
        float *patch = new float[64]; std::memcpy(patch, obs_patch + (size_t)k * 64, 256);
        Feature *f = new Feature(pt, patch, V2D(obs_px[2 * k], obs_px[2 * k + 1]), V3D(obs_f[3 * k], obs_f[3 * k + 1], obs_f[3 * k + 2]), SE3(R, V3D(obs_t[3 * k], obs_t[3 * k + 1], obs_t[3 * k + 2])),
                                 obs_level[k]);
        f->img_ = refs[obs_img_idx[k]]; f->id_ = obs_id[k]; f->inv_expo_time_ = obs_inv_expo[k];
        pt->obs_.push_back(f);                            // list order = the order given
        obs_index[f] = k;
        if (ref_patch_in[i] == k) { pt->ref_patch = f; pt->has_ref_patch_ = true; }
      }
    }
    pts[i] = pt; pt_index[pt] = i;
    VOXEL_LOCATION key(keys[3 * i], keys[3 * i + 1], keys[3 * i + 2]);
    auto it = vio.feat_map.find(key);
    if (it == vio.feat_map.end()) { VOXEL_POINTS *vp = new VOXEL_POINTS(0); vio.feat_map[key] = vp; it = vio.feat_map.find(key); }
    it->second->voxel_points.push_back(pt); it->second->count++;
  }
  std::vector<pointWithVar> pgv(n_pg);
  for (int i = 0; i < n_pg; i++) pgv[i].point_w = V3D(pg[3 * i], pg[3 * i + 1], pg[3 * i + 2]);
  std::unordered_map<VOXEL_LOCATION, VoxelOctoTree *> plane_map;
  if (g_rc.map) plane_map = ((MapHandle *)g_rc.map)->voxel_map;              // (a copy of the key -> tree pointers; the trees stay the MapHandle's)
  vio.resetGrid();
  { CoutCapture cap; vio.retrieveFromVisualSparseMap(cur, pgv, plane_map); }
  if (g_rc.n_add) {
    const auto &add = vio.visual_submap->add_from_voxel_map;
    *g_rc.n_add = (int32_t)add.size();
    for (int k = 0; k < (int)add.size() && k < g_rc.cap && g_rc.add6; k++)
      for (int j = 0; j < 3; j++) { g_rc.add6[6 * k + j] = add[k].point_w[j]; g_rc.add6[6 * k + 3 + j] = add[k].normal[j]; }
  }
  g_rc = RaycastSetup{};
  for (int i = 0; i < vio.length; i++) {
    cell_type[i] = vio.grid_num[i]; cell_dist[i] = vio.map_dist[i];
    cell_point[i] = (vio.grid_num[i] == VIOManager::TYPE_MAP && vio.retrieve_voxel_points[i]) ? pt_index[vio.retrieve_voxel_points[i]] : -1;
  }
  for (int i = 0; i < n_pts; i++) ref_patch_out[i] = (pts[i]->has_ref_patch_ && pts[i]->ref_patch) ? obs_index[pts[i]->ref_patch] : -1;
  SubSparseMap &sm = *vio.visual_submap;
  const int L = c->patch_pyrimid_level;
  *n_sub = (int)sm.voxel_points.size();
  for (int k = 0; k < *n_sub; k++) {
    sub_point[k] = pt_index[sm.voxel_points[k]];
    sub_obs[k] = sm.voxel_points[k]->ref_patch ? obs_index[sm.voxel_points[k]->ref_patch] : -1;
    sub_search[k] = sm.search_levels[k]; sub_error[k] = sm.errors[k]; sub_inv_expo[k] = sm.inv_expo_list[k];
    std::memcpy(sub_patch + (size_t)k * L * 64, sm.warp_patch[k].data(), (size_t)L * 256);
  }
  return vio.length;
}

// calcBodyCov (voxel_map.cpp:15-34)
void orc_calc_body_cov(const double *pb3, float range_inc, float degree_inc, double deg2rad, double *cov9, double *pb_out3) {
  (void)deg2rad;
  Eigen::Vector3d pb(pb3[0], pb3[1], pb3[2]); Eigen::Matrix3d cov;
  calcBodyCov(pb, range_inc, degree_inc, cov);
  rm_out(cov, cov9);
  if (pb_out3) for (int k = 0; k < 3; k++) pb_out3[k] = pb[k];
}

// VoxelOctoTree::init_plane (voxel_map.cpp:55-135)
int orc_init_plane(const double *point_w, const double *var9, int n, float planer_threshold, PlaneFitPOD *out) {
  VoxelOctoTree node(2, 0, 5, 50, planer_threshold);
  std::vector<pointWithVar> pts; to_points(point_w, var9, n, pts);
  node.init_plane(pts, node.plane_ptr_);
  const VoxelPlane &p = *node.plane_ptr_;
  std::memset(out, 0, sizeof(*out));
  for (int k = 0; k < 3; k++) { out->center[k] = p.center_[k]; out->normal[k] = p.normal_[k]; }
  if (p.is_plane_) for (int k = 0; k < 3; k++) { out->y_normal[k] = p.y_normal_[k]; out->x_normal[k] = p.x_normal_[k]; }
  rm_out(p.covariance_, out->covariance); rm_out(p.plane_var_, out->plane_var);
  out->radius = p.radius_; out->min_eigen_value = p.min_eigen_value_; out->mid_eigen_value = p.mid_eigen_value_; out->max_eigen_value = p.max_eigen_value_; out->d = p.d_;
  out->points_size = p.points_size_; out->is_plane = p.is_plane_ ? 1 : 0;
  return 0;
}

// StatesGroup boxplus / boxminus (common_lib.h:170-206), Exp / Log (so3_math.h:44-66)
void orc_state_boxplus(const StatePOD *s, const double *d19, StatePOD *out) { StatesGroup g; from_pod(g, *s); Eigen::Matrix<double, DIM_STATE, 1> d; for (int i = 0; i < 19; i++) d(i) = d19[i]; g += d; to_pod(g, *out); }
void orc_state_boxminus(const StatePOD *a, const StatePOD *b, double *out19) { StatesGroup ga, gb; from_pod(ga, *a); from_pod(gb, *b); auto d = ga - gb; for (int i = 0; i < 19; i++) out19[i] = d(i); }
void orc_so3_exp(const double *v3, double *R9) { M3D R = Exp(v3[0], v3[1], v3[2]); rm_out(R, R9); }
void orc_so3_log(const double *R9, double *v3) { M3D R; rm_in(R, R9); V3D v = Log(R); for (int k = 0; k < 3; k++) v3[k] = v[k]; }

} // extern "C"

