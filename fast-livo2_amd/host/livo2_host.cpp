// See livo2_host.hpp.  Pure data movement between the reference-shaped containers and the C ABI.
#include "livo2_host.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>

namespace livo2 {

namespace {
struct Flat {
  std::vector<int64_t> root_key; std::vector<int32_t> root_node; std::vector<double> root_center; std::vector<float> root_quarter;
  std::vector<int32_t> node_plane, node_child;
  std::vector<double> plane_normal, plane_center, plane_var; std::vector<float> plane_d, plane_radius;
};

int32_t flatten_node(const VoxelOctoTree *t, int layer, Flat &f, std::unordered_map<const VoxelPlane *, int32_t> &pidx, std::vector<const VoxelPlane *> &by_index,
                     std::vector<int> &plane_layer) {
  const int32_t me = (int32_t)f.node_plane.size();
  f.node_plane.push_back(-1);
  f.node_child.insert(f.node_child.end(), 8, -1);
  const VoxelPlane *p = t->plane_ptr_;
  if (p && p->is_plane_) {
    const int32_t pi = (int32_t)f.plane_d.size();
    pidx[p] = pi; by_index.push_back(p); plane_layer.push_back(layer);
    f.plane_normal.insert(f.plane_normal.end(), p->normal_.begin(), p->normal_.end());
    f.plane_center.insert(f.plane_center.end(), p->center_.begin(), p->center_.end());
    f.plane_var.insert(f.plane_var.end(), p->plane_var_.begin(), p->plane_var_.end());
    f.plane_d.push_back(p->d_); f.plane_radius.push_back(p->radius_);
    f.node_plane[me] = pi;
  }
  for (int k = 0; k < 8; k++)
    if (t->leaves_[k]) { const int32_t c = flatten_node(t->leaves_[k], layer + 1, f, pidx, by_index, plane_layer); f.node_child[(size_t)me * 8 + k] = c; }
  return me;
}
} // namespace

void VoxelMapManager::FlattenAndUpload() {
  Flat f;
  plane_index_.clear(); plane_by_index_.clear(); plane_layer_.clear();
  for (const auto &kv : voxel_map_) {
    f.root_key.push_back(kv.first.x); f.root_key.push_back(kv.first.y); f.root_key.push_back(kv.first.z);
    for (int k = 0; k < 3; k++) f.root_center.push_back(kv.second->voxel_center_[k]);
    f.root_quarter.push_back(kv.second->quater_length_);
    f.root_node.push_back(flatten_node(kv.second, 0, f, plane_index_, plane_by_index_, plane_layer_));
  }
  livo2_map_view mv{};
  mv.n_roots = (int32_t)f.root_node.size(); mv.n_nodes = (int32_t)f.node_plane.size(); mv.n_planes = (int32_t)f.plane_d.size();
  mv.root_key = f.root_key.data(); mv.root_node = f.root_node.data(); mv.root_center = f.root_center.data(); mv.root_quarter = f.root_quarter.data();
  mv.node_plane = f.node_plane.data(); mv.node_child = f.node_child.data();
  mv.plane_normal = f.plane_normal.data(); mv.plane_center = f.plane_center.data(); mv.plane_var = f.plane_var.data();
  mv.plane_d = f.plane_d.data(); mv.plane_radius = f.plane_radius.data();
  dev_.check(livo2_map_upload(dev_.ctx(), &mv));
  map_dirty_ = false;
}

void VoxelMapManager::RefreshPlanes(const std::vector<const VoxelPlane *> &planes) {
  if (map_dirty_) return;                   // a full upload is pending anyway
  std::vector<int32_t> idx; std::vector<double> n, c, pv; std::vector<float> d, r;
  for (const VoxelPlane *p : planes) {
    auto it = plane_index_.find(p);
    if (it == plane_index_.end() || !p->is_plane_) { map_dirty_ = true; return; }   // structural change: re-flatten
    idx.push_back(it->second);
    n.insert(n.end(), p->normal_.begin(), p->normal_.end()); c.insert(c.end(), p->center_.begin(), p->center_.end());
    pv.insert(pv.end(), p->plane_var_.begin(), p->plane_var_.end()); d.push_back(p->d_); r.push_back(p->radius_);
  }
  dev_.check(livo2_map_update_planes(dev_.ctx(), idx.data(), (int32_t)idx.size(), n.data(), c.data(), pv.data(), d.data(), r.data()));
}

// ---- map maintenance with device-side plane fits ---------------------------------------------------------------------------------------
VoxelOctoTree *VoxelOctoTree::clone() const {
  VoxelOctoTree *c = new VoxelOctoTree(max_layer_, layer_, points_size_threshold_, max_points_num_, planer_threshold_);
  c->temp_points_ = temp_points_; *c->plane_ptr_ = *plane_ptr_;
  std::memcpy(c->voxel_center_, voxel_center_, sizeof(voxel_center_)); c->quater_length_ = quater_length_;
  c->octo_state_ = octo_state_; c->new_points_ = new_points_; c->update_size_threshold_ = update_size_threshold_;
  c->init_octo_ = init_octo_; c->update_enable_ = update_enable_; c->layer_init_num_ = layer_init_num_;
  for (int k = 0; k < 8; k++) c->leaves_[k] = leaves_[k] ? leaves_[k]->clone() : nullptr;
  return c;
}

namespace {

typedef std::vector<uint8_t> NodePath;                       // child indices from the root
struct FitRequest { NodePath path; int seq; std::vector<pointWithVar> points; };
typedef std::map<std::pair<NodePath, int>, livo2_plane_fit> FitCache;

// One replay of a root voxel's share of the scan on a scratch copy of its subtree.  The octree logic below is the reference's
// (UpdateOctoTree / init_octo_tree / cut_octo_tree, src/voxel_map.cpp:137-290) with init_plane replaced by a lookup of the device's
// answer; a missing answer is queued and the replay abandoned at the next point where the control flow would depend on it.
struct Replay {
  const FitCache &cache;
  std::vector<FitRequest> requests;
  std::map<NodePath, int> calls;                             // init_plane calls seen per node in this replay
  bool pending = false;
  explicit Replay(const FitCache &c) : cache(c) {}

  bool fit(VoxelOctoTree *node, const NodePath &path) {      // VoxelOctoTree::init_plane(node->temp_points_, node->plane_ptr_)
    const int seq = calls[path]++;
    auto it = cache.find({path, seq});
    if (it == cache.end()) { requests.push_back({path, seq, node->temp_points_}); pending = true; return false; }
    const livo2_plane_fit &f = it->second;
    VoxelPlane *p = node->plane_ptr_;
    std::memcpy(p->center_.data(), f.center, 24); std::memcpy(p->covariance_.data(), f.covariance, 72); std::memcpy(p->plane_var_.data(), f.plane_var, 288);
    std::memcpy(p->normal_.data(), f.normal, 24); p->points_size_ = f.points_size; p->radius_ = f.radius;
    if (f.is_plane) {
      std::memcpy(p->y_normal_.data(), f.y_normal, 24); std::memcpy(p->x_normal_.data(), f.x_normal, 24);
      p->min_eigen_value_ = f.min_eigen_value; p->mid_eigen_value_ = f.mid_eigen_value; p->max_eigen_value_ = f.max_eigen_value; p->d_ = f.d;
    }
    p->is_plane_ = f.is_plane != 0; p->is_update_ = true;
    return true;
  }

  VoxelOctoTree *new_leaf(VoxelOctoTree *n, const int xyz[3]) {          // voxel_map.cpp:179-186 / 255-262
    VoxelOctoTree *l = new VoxelOctoTree(n->max_layer_, n->layer_ + 1, n->layer_init_num_[n->layer_ + 1], n->max_points_num_, n->planer_threshold_);
    l->layer_init_num_ = n->layer_init_num_;
    for (int k = 0; k < 3; k++) l->voxel_center_[k] = n->voxel_center_[k] + (2 * xyz[k] - 1) * n->quater_length_;
    l->quater_length_ = n->quater_length_ / 2;
    return l;
  }
  static int leaf_of(const VoxelOctoTree *n, const pointWithVar &pv, int xyz[3]) {
    for (int k = 0; k < 3; k++) xyz[k] = pv.point_w[k] > n->voxel_center_[k] ? 1 : 0;
    return 4 * xyz[0] + 2 * xyz[1] + xyz[2];
  }

  void cut_octo_tree(VoxelOctoTree *n, const NodePath &path) {           // voxel_map.cpp:163-217
    if (n->layer_ >= n->max_layer_) { n->octo_state_ = 0; return; }
    for (const pointWithVar &pv : n->temp_points_) {
      int xyz[3]; const int leafnum = leaf_of(n, pv, xyz);
      if (!n->leaves_[leafnum]) n->leaves_[leafnum] = new_leaf(n, xyz);
      n->leaves_[leafnum]->temp_points_.push_back(pv);
      n->leaves_[leafnum]->new_points_++;
    }
    for (int i = 0; i < 8; i++) {
      VoxelOctoTree *l = n->leaves_[i];
      if (!l || (int)l->temp_points_.size() <= l->points_size_threshold_) continue;
      NodePath lp = path; lp.push_back((uint8_t)i);
      if (!fit(l, lp)) continue;                                           // the siblings' fits do not depend on this one: ask for all of them in one sweep
      if (l->plane_ptr_->is_plane_) {
        l->octo_state_ = 0;
        if ((int)l->temp_points_.size() > l->max_points_num_) { l->update_enable_ = false; std::vector<pointWithVar>().swap(l->temp_points_); n->new_points_ = 0; }
      } else { l->octo_state_ = 1; cut_octo_tree(l, lp); }
      l->init_octo_ = true;
      l->new_points_ = 0;
    }
  }

  void init_octo_tree(VoxelOctoTree *n, const NodePath &path) {          // voxel_map.cpp:137-161
    if ((int)n->temp_points_.size() <= n->points_size_threshold_) return;
    if (!fit(n, path)) return;
    if (n->plane_ptr_->is_plane_) {
      n->octo_state_ = 0;
      if ((int)n->temp_points_.size() > n->max_points_num_) { n->update_enable_ = false; std::vector<pointWithVar>().swap(n->temp_points_); n->new_points_ = 0; }
    } else { n->octo_state_ = 1; cut_octo_tree(n, path); }
    n->init_octo_ = true;
    n->new_points_ = 0;
  }

  void UpdateOctoTree(VoxelOctoTree *n, const NodePath &path, const pointWithVar &pv) {   // voxel_map.cpp:219-290
    if (!n->init_octo_) {
      n->new_points_++; n->temp_points_.push_back(pv);
      if ((int)n->temp_points_.size() > n->points_size_threshold_) init_octo_tree(n, path);
      return;
    }
    if (n->plane_ptr_->is_plane_) {
      if (n->update_enable_) {
        n->new_points_++; n->temp_points_.push_back(pv);
        if (n->new_points_ > n->update_size_threshold_) { if (!fit(n, path)) return; n->new_points_ = 0; }
        if ((int)n->temp_points_.size() >= n->max_points_num_) { n->update_enable_ = false; std::vector<pointWithVar>().swap(n->temp_points_); n->new_points_ = 0; }
      }
      return;
    }
    if (n->layer_ < n->max_layer_) {
      int xyz[3]; const int leafnum = leaf_of(n, pv, xyz);
      if (!n->leaves_[leafnum]) n->leaves_[leafnum] = new_leaf(n, xyz);
      NodePath lp = path; lp.push_back((uint8_t)leafnum);
      UpdateOctoTree(n->leaves_[leafnum], lp, pv);
    } else if (n->update_enable_) {
      n->new_points_++; n->temp_points_.push_back(pv);
      if (n->new_points_ > n->update_size_threshold_) { if (!fit(n, path)) return; n->new_points_ = 0; }
      if ((int)n->temp_points_.size() > n->max_points_num_) { n->update_enable_ = false; std::vector<pointWithVar>().swap(n->temp_points_); n->new_points_ = 0; }
    }
  }
};

struct RootWork { VOXEL_LOCATION key; std::vector<int> points; FitCache cache; bool done = false; };

} // namespace

void VoxelMapManager::maintain(const std::vector<pointWithVar> &input_points, bool build) {
  const float voxel_size = (float)config_setting_.max_voxel_size_;
  const float planer_threshold = (float)config_setting_.planner_threshold_;
  // root voxel of every point, in input order (voxel_map.cpp:559-567 / 617-625: float division, -1 for negatives, truncation)
  std::unordered_map<VOXEL_LOCATION, size_t, VoxelLocationHash> index;
  std::vector<RootWork> work;
  for (size_t i = 0; i < input_points.size(); i++) {
    float loc_xyz[3];
    for (int j = 0; j < 3; j++) { loc_xyz[j] = (float)(input_points[i].point_w[j] / voxel_size); if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0; }
    const VOXEL_LOCATION position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
    auto it = index.find(position);
    if (it == index.end()) { it = index.emplace(position, work.size()).first; work.push_back(RootWork{position, {}, {}, false}); }
    work[it->second].points.push_back((int)i);
  }
  last_fit_rounds_ = 0; last_fit_count_ = 0;
  size_t remaining = work.size();
  while (remaining > 0) {
    std::vector<double> pw, var; std::vector<int32_t> off{0};
    struct Owner { size_t w; NodePath path; int seq; };
    std::vector<Owner> owners;
    for (size_t w = 0; w < work.size(); w++) {
      RootWork &rw = work[w];
      if (rw.done) continue;
      auto it = voxel_map_.find(rw.key);
      VoxelOctoTree *root;
      if (it != voxel_map_.end()) root = it->second->clone();
      else {                                                             // voxel_map.cpp:574-583 / 630-637
        root = new VoxelOctoTree(config_setting_.max_layer_, 0, config_setting_.layer_init_num_[0], config_setting_.max_points_num_, planer_threshold);
        root->layer_init_num_ = config_setting_.layer_init_num_;
        root->quater_length_ = voxel_size / 4;
        root->voxel_center_[0] = (0.5 + rw.key.x) * voxel_size; root->voxel_center_[1] = (0.5 + rw.key.y) * voxel_size; root->voxel_center_[2] = (0.5 + rw.key.z) * voxel_size;
      }
      Replay rp(rw.cache);
      if (build) {                                                       // BuildVoxelMap: all points first, then init_octo_tree (voxel_map.cpp:568-590)
        for (int i : rw.points) { root->temp_points_.push_back(input_points[i]); root->new_points_++; }
        rp.init_octo_tree(root, NodePath());
      } else {
        for (int i : rw.points) { rp.UpdateOctoTree(root, NodePath(), input_points[i]); if (rp.pending) break; }
      }
      if (!rp.pending) {
        if (it != voxel_map_.end()) { delete it->second; it->second = root; } else voxel_map_[rw.key] = root;
        rw.done = true; remaining--;
        continue;
      }
      delete root;
      for (FitRequest &rq : rp.requests) {
        for (const pointWithVar &pv : rq.points) { pw.insert(pw.end(), pv.point_w.begin(), pv.point_w.end()); var.insert(var.end(), pv.var.begin(), pv.var.end()); }
        off.push_back((int32_t)(pw.size() / 3));
        owners.push_back({w, rq.path, rq.seq});
      }
    }
    if (owners.empty()) break;
    std::vector<livo2_plane_fit> fit(owners.size());
    dev_.check(livo2_plane_fit_batch(dev_.ctx(), pw.data(), var.data(), off.data(), (int32_t)owners.size(), planer_threshold, nullptr, fit.data()));
    for (size_t g = 0; g < owners.size(); g++) work[owners[g].w].cache[{owners[g].path, owners[g].seq}] = fit[g];
    last_fit_rounds_++; last_fit_count_ += (int)owners.size();
  }
  map_dirty_ = true;                                                      // the snapshot is re-flattened before the next StateEstimation
}

static void device_tree_feed(Device &dev, const std::vector<pointWithVar> &pts, bool build) {
  std::vector<double> pw(pts.size() * 3), var(pts.size() * 9);
  for (size_t i = 0; i < pts.size(); i++) { std::memcpy(&pw[i * 3], pts[i].point_w.data(), 24); std::memcpy(&var[i * 9], pts[i].var.data(), 72); }
  dev.check(livo2_map_tree_update(dev.ctx(), pw.data(), var.data(), (int32_t)pts.size(), build ? 1 : 0));
}
void VoxelMapManager::BuildVoxelMap(const std::vector<pointWithVar> &input_points) {
  if (!device_map_) { maintain(input_points, true); return; }
  livo2_map_tree_cfg tc{};
  tc.voxel_size = config_setting_.max_voxel_size_; tc.planer_threshold = config_setting_.planner_threshold_; tc.max_layer = config_setting_.max_layer_;
  tc.max_points_num = config_setting_.max_points_num_; tc.max_roots = device_map_max_roots_;
  for (int k = 0; k < 5; k++) tc.layer_init_num[k] = config_setting_.layer_init_num_[std::min<size_t>(k, config_setting_.layer_init_num_.size() - 1)];
  dev_.check(livo2_map_tree_create(dev_.ctx(), &tc));
  device_tree_feed(dev_, input_points, true);
  last_map_kernel_us_ = livo2_map_tree_last_kernel_us(dev_.ctx());
  map_dirty_ = false;
}
void VoxelMapManager::UpdateVoxelMap(const std::vector<pointWithVar> &input_points) {
  if (!device_map_) { maintain(input_points, false); return; }
  device_tree_feed(dev_, input_points, false);
  last_map_kernel_us_ = livo2_map_tree_last_kernel_us(dev_.ctx());
}
void VoxelMapManager::UpdateVoxelMapFromPosterior() {
  if (!device_map_) throw std::runtime_error("UpdateVoxelMapFromPosterior needs device_map_");
  livo2_lidar_cfg cfg{};
  cfg.max_iterations = config_setting_.max_iterations_; cfg.max_layer = config_setting_.max_layer_; cfg.sigma_num = config_setting_.sigma_num_;
  cfg.dept_err = config_setting_.dept_err_; cfg.beam_err = config_setting_.beam_err_; cfg.voxel_size = config_setting_.max_voxel_size_; cfg.deg2rad = 0.0;
  std::memcpy(cfg.extR, extR_.data(), 72); std::memcpy(cfg.extT, extT_.data(), 24);
  livo2_state s; state_.to_abi(s);
  if (async_map_update_) {
    // state_ untouched since StateEstimation wrote it (LIVMapper.cpp:371-413 only reads it): the posterior is still on the device, no upload
    const bool resident = posterior_valid_ && std::memcmp(&s, &posterior_, sizeof(s)) == 0;
    dev_.check(livo2_map_tree_update_from_scan_async(dev_.ctx(), resident ? nullptr : &s, &cfg));
    if (!host_point_lists_) return;
    // (host lists wanted: pv_list_ at the posterior is on the context's stream already — it is read back below while the octree update runs on the second stream)
  } else {
    dev_.check(livo2_map_tree_update_from_scan(dev_.ctx(), &s, &cfg, 0));
    last_map_kernel_us_ = livo2_map_tree_last_kernel_us(dev_.ctx());
  }
  if (host_point_lists_ && !pv_list_.empty()) {        // the reference leaves the posterior point_w / var in pv_list_ (LIVMapper.cpp:417-424; `_pv_list` of :426 feeds handleVIO)
    const int32_t n = (int32_t)pv_list_.size();
    const size_t bytes = (size_t)n * 96 + 64;
    if (bytes > pin_bytes_) {
      livo2_host_free_pinned(pin_); pin_ = nullptr; pin_bytes_ = 0;
      dev_.check(livo2_host_alloc_pinned(bytes + bytes / 4, &pin_));
      pin_bytes_ = bytes + bytes / 4;
    }
    double *pw = static_cast<double *>(pin_), *var = pw + (size_t)n * 3;             // the per-point receive buffer of StateEstimation is free again by now
    int32_t got = 0;
    dev_.check(livo2_map_tree_read_pv(dev_.ctx(), pw, var, n, &got));
    const int threads = std::max(1, std::min(fill_threads_, got / 2048 + 1));
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int32_t i = 0; i < got; i++) { std::memcpy(pv_list_[i].point_w.data(), &pw[(size_t)i * 3], 24); std::memcpy(pv_list_[i].var.data(), &var[(size_t)i * 9], 72); }
  }
}

void VoxelMapManager::JoinMapUpdate() {
  if (!device_map_) return;
  dev_.check(livo2_map_tree_update_join(dev_.ctx()));
  last_map_kernel_us_ = livo2_map_tree_last_kernel_us(dev_.ctx());
}

int VoxelMapManager::mapSliding() {
  if (device_map_) {
    int32_t removed = 0;
    dev_.check(livo2_map_tree_slide(dev_.ctx(), position_last_.data(), config_setting_.sliding_thresh, config_setting_.half_map_size, &removed, nullptr));
    return removed;
  }
  const double dx = position_last_[0] - last_slide_position[0], dy = position_last_[1] - last_slide_position[1], dz = position_last_[2] - last_slide_position[2];
  if (std::sqrt((dx * dx + dy * dy) + dz * dz) < config_setting_.sliding_thresh) return -1;
  last_slide_position = position_last_;
  float loc_xyz[3];
  for (int j = 0; j < 3; j++) {
    loc_xyz[j] = position_last_[j] / config_setting_.max_voxel_size_;
    if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
  }
  const int h = config_setting_.half_map_size;
  const int x_max = (int)((int64_t)loc_xyz[0] + h), x_min = (int)((int64_t)loc_xyz[0] - h), y_max = (int)((int64_t)loc_xyz[1] + h), y_min = (int)((int64_t)loc_xyz[1] - h);
  const int z_max = (int)((int64_t)loc_xyz[2] + h), z_min = (int)((int64_t)loc_xyz[2] - h);
  int n = 0;
  for (auto it = voxel_map_.begin(); it != voxel_map_.end();) {
    const VOXEL_LOCATION &loc = it->first;
    if (loc.x > x_max || loc.x < x_min || loc.y > y_max || loc.y < y_min || loc.z > z_max || loc.z < z_min) { delete it->second; it = voxel_map_.erase(it); n++; }
    else ++it;
  }
  if (n) map_dirty_ = true;                                    // the resident snapshot no longer matches voxel_map_
  return n;
}

void VoxelMapManager::FitPlanes(const std::vector<VoxelOctoTree *> &voxels) {
  const int G = (int)voxels.size();
  if (G == 0) return;
  std::vector<double> pw, var; std::vector<int32_t> off{0}, pidx(G, -1);
  for (int g = 0; g < G; g++) {
    for (const pointWithVar &pv : voxels[g]->temp_points_) { pw.insert(pw.end(), pv.point_w.begin(), pv.point_w.end()); var.insert(var.end(), pv.var.begin(), pv.var.end()); }
    off.push_back((int32_t)(pw.size() / 3));
    if (!map_dirty_) { auto it = plane_index_.find(voxels[g]->plane_ptr_); if (it != plane_index_.end()) pidx[g] = it->second; }
  }
  std::vector<livo2_plane_fit> fit(G);
  // all voxels of one tree share planer_threshold_ (reference include/voxel_map.h:147)
  dev_.check(livo2_plane_fit_batch(dev_.ctx(), pw.data(), var.data(), off.data(), G, voxels[0]->planer_threshold_, map_dirty_ ? nullptr : pidx.data(), fit.data()));
  for (int g = 0; g < G; g++) {
    VoxelPlane *p = voxels[g]->plane_ptr_;
    const livo2_plane_fit &f = fit[g];
    const bool was_plane = p->is_plane_;
    std::memcpy(p->center_.data(), f.center, 24); std::memcpy(p->covariance_.data(), f.covariance, 72); p->points_size_ = f.points_size;
    std::memcpy(p->plane_var_.data(), f.plane_var, 288); std::memcpy(p->normal_.data(), f.normal, 24); p->radius_ = f.radius;
    if (f.is_plane) {
      std::memcpy(p->y_normal_.data(), f.y_normal, 24); std::memcpy(p->x_normal_.data(), f.x_normal, 24);
      p->min_eigen_value_ = f.min_eigen_value; p->mid_eigen_value_ = f.mid_eigen_value; p->max_eigen_value_ = f.max_eigen_value; p->d_ = f.d;
    }
    p->is_plane_ = f.is_plane != 0; p->is_update_ = true;
    if (p->is_plane_ != was_plane || (p->is_plane_ && pidx[g] < 0)) map_dirty_ = true;      // the snapshot's shape changed: re-flatten before the next update
  }
}

void VoxelMapManager::StateEstimation(StatesGroup &state_propagat) {
  if (!device_map_ && map_dirty_) FlattenAndUpload();
  const int n = (int)feats_down_body_.size();
  feats_down_size_ = n;
  livo2_lidar_cfg cfg{};
  cfg.max_iterations = config_setting_.max_iterations_; cfg.max_layer = config_setting_.max_layer_; cfg.sigma_num = config_setting_.sigma_num_;
  cfg.dept_err = config_setting_.dept_err_; cfg.beam_err = config_setting_.beam_err_; cfg.voxel_size = config_setting_.max_voxel_size_; cfg.deg2rad = 0.0;
  std::memcpy(cfg.extR, extR_.data(), 72); std::memcpy(cfg.extT, extT_.data(), 24);
  static_assert(sizeof(PointXYZ) == 12, "xyz AoS");
  static const bool shim_prof = std::getenv("LIVO2_SHIM_PROF") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
  if (!scan_resident_) dev_.check(livo2_lidar_set_scan(dev_.ctx(), n ? &feats_down_body_[0].x : nullptr, n, &cfg));
  scan_resident_ = false;
  const double t_scan = since();

  // per-point outputs land in page-locked buffers the manager keeps (D2H at PCIe speed, no bounce through the runtime's staging)
  int32_t *match = nullptr, *normal_plane = nullptr; float *dis = nullptr, *pw = nullptr; double *var = nullptr, *bcov = nullptr;
  livo2_lidar_points pts{};
  if (host_point_lists_) {
    const size_t nn = (size_t)std::max(n, 1);
    const size_t bytes = nn * (4 + 4 + 4 + 12 + 72 + 72) + 6 * 64;
    if (bytes > pin_bytes_) {
      livo2_host_free_pinned(pin_); pin_ = nullptr; pin_bytes_ = 0;
      dev_.check(livo2_host_alloc_pinned(bytes + bytes / 4, &pin_));
      pin_bytes_ = bytes + bytes / 4;
    }
    char *b = static_cast<char *>(pin_);
    auto take = [&](size_t sz) { char *q = b; b += (sz + 63) & ~(size_t)63; return q; };
    var = reinterpret_cast<double *>(take(nn * 72)); bcov = reinterpret_cast<double *>(take(nn * 72)); pw = reinterpret_cast<float *>(take(nn * 12));
    match = reinterpret_cast<int32_t *>(take(nn * 4)); normal_plane = reinterpret_cast<int32_t *>(take(nn * 4)); dis = reinterpret_cast<float *>(take(nn * 4));
    pts.match_plane = match; pts.dis_to_plane = dis; pts.point_w = pw; pts.normal_plane = normal_plane; pts.var = var; pts.body_cov = bcov; pts.pinned = 1;
  }
  livo2_state s_in, s_prop;
  state_.to_abi(s_in); state_propagat.to_abi(s_prop);
  livo2_lidar_result res;
  dev_.check(livo2_lidar_update(dev_.ctx(), &s_in, &s_prop, &cfg, &res, host_point_lists_ ? &pts : nullptr));
  const double t_update = since();
  state_.from_abi(res.state);
  state_.to_abi(posterior_); posterior_valid_ = true;                              // (what UpdateVoxelMapFromPosterior compares state_ with)
  std::memcpy(position_last_.data(), res.position_last, 24);
  {   // euler_cur = RotMtoEuler(state_.rot_end) (so3_math.h:68-87); geoQuat_ = tf::createQuaternionMsgFromRollPitchYaw(euler_cur) (voxel_map.cpp:493)
    const double *R = res.state.rot;
    const double sy = std::sqrt(R[0] * R[0] + R[3] * R[3]);
    double ex, ey, ez;
    if (!(sy < 1e-6)) { ex = std::atan2(R[7], R[8]); ey = std::atan2(-R[6], sy); ez = std::atan2(R[3], R[0]); }
    else { ex = std::atan2(-R[5], R[4]); ey = std::atan2(-R[6], sy); ez = 0.0; }
    const double hr = ex * 0.5, hp = ey * 0.5, hy = ez * 0.5;                       // tf::Quaternion::setRPY
    const double cr = std::cos(hr), sr = std::sin(hr), cp = std::cos(hp), sp = std::sin(hp), cy = std::cos(hy), sn = std::sin(hy);
    geoQuat_ = {{sr * cp * cy - cr * sp * sn, cr * sp * cy + sr * cp * sn, cr * cp * sn - sr * sp * cy, cr * cp * cy + sr * sp * sn}};
  }

  if (!host_point_lists_) {          // see livo2_host.hpp: the per-point members stay empty
    pv_list_.clear(); ptpl_list_.clear(); cross_mat_list_.clear(); body_cov_list_.clear();
    effct_feat_num_ = res.n_iters > 0 ? res.iter_sums[res.n_iters - 1].n_eff : 0;
    if (shim_prof) std::fprintf(stderr, "StateEstimation (lean, %d points): set_scan %.3f ms, update %.3f ms\n", n, t_scan, t_update - t_scan);
    return;
  }
  // what the reference leaves behind for LIVMapper (src/LIVMapper.cpp:371-426, 446) and VIO (src/vio.cpp:811): ~900 B of reference structs per point, written by
  // fill_threads_ host threads (the lists are indexed, so the points are independent once the position of every match in ptpl_list_ is known)
  // device-resident map: the VoxelPlane members of the matched rows come from the device plane table
  std::vector<int32_t> rows, row_pos; std::vector<double> r_normal, r_center, r_pvar; std::vector<float> r_d; std::vector<int32_t> r_layer;
  if (device_map_) {
    int32_t max_row = -1;
    for (int i = 0; i < n; i++) max_row = std::max(max_row, std::max(match[i], normal_plane[i]));
    row_pos.assign((size_t)(max_row + 1), -1);
    for (int i = 0; i < n; i++)
      for (int32_t r : {match[i], normal_plane[i]}) if (r >= 0 && row_pos[r] < 0) { row_pos[r] = (int32_t)rows.size(); rows.push_back(r); }
    r_normal.resize(rows.size() * 3); r_center.resize(rows.size() * 3); r_pvar.resize(rows.size() * 36); r_d.resize(rows.size()); r_layer.resize(rows.size());
    dev_.check(livo2_map_tree_read_planes(dev_.ctx(), rows.data(), (int32_t)rows.size(), r_normal.data(), r_center.data(), r_pvar.data(), r_d.data(), nullptr, r_layer.data()));
  }
  const double t_planes = since();
  std::vector<int32_t> slot((size_t)n);                                     // position of point i's PointToPlane in ptpl_list_ (the reference pushes them in scan order)
  int n_match = 0;
  for (int i = 0; i < n; i++) { slot[i] = n_match; n_match += match[i] >= 0 ? 1 : 0; }
  pv_list_.resize(n); cross_mat_list_.resize(n); body_cov_list_.resize(n); ptpl_list_.resize(n_match);
  const int threads = std::max(1, std::min(fill_threads_, n / 2048 + 1));
#pragma omp parallel for schedule(static) num_threads(threads)
  for (int i = 0; i < n; i++) {
    pointWithVar &pv = pv_list_[i];
    const PointXYZ &p = feats_down_body_[i];
    pv.point_b = {p.x, p.y, p.z};
    pv.point_i = {}; pv.var_nostate = {}; pv.point_crossmat = {}; pv.normal = {};
    pv.point_w = {pw[(size_t)i * 3], pw[(size_t)i * 3 + 1], pw[(size_t)i * 3 + 2]};
    std::memcpy(pv.var.data(), &var[(size_t)i * 9], 72);
    std::memcpy(pv.body_var.data(), &bcov[(size_t)i * 9], 72);
    body_cov_list_[i] = pv.body_var;
    double pz = (p.z == 0) ? 0.001 : (double)p.z;                                   // reference src/voxel_map.cpp:352-358
    double q[3];
    for (int j = 0; j < 3; j++) q[j] = extR_[j * 3] * p.x + extR_[j * 3 + 1] * p.y + extR_[j * 3 + 2] * pz + extT_[j];
    cross_mat_list_[i] = {0.0, -q[2], q[1], q[2], 0.0, -q[0], -q[1], q[0], 0.0};
    if (device_map_) {
      if (normal_plane[i] >= 0) std::memcpy(pv.normal.data(), &r_normal[(size_t)row_pos[normal_plane[i]] * 3], 24);
      if (match[i] >= 0) {
        const size_t k = (size_t)row_pos[match[i]];
        PointToPlane &pp = ptpl_list_[slot[i]];
        pp.point_b_ = pv.point_b; pp.point_w_ = pv.point_w; std::memcpy(pp.normal_.data(), &r_normal[k * 3], 24); std::memcpy(pp.center_.data(), &r_center[k * 3], 24);
        std::memcpy(pp.plane_var_.data(), &r_pvar[k * 36], 288);
        pp.body_cov_ = pv.body_var; pp.layer_ = r_layer[k]; pp.d_ = r_d[k]; pp.eigen_value_ = 0; pp.is_valid_ = true; pp.dis_to_plane_ = dis[i];
      }
      continue;
    }
    if (normal_plane[i] >= 0) pv.normal = plane_by_index_[normal_plane[i]]->normal_;
    if (match[i] >= 0) {
      const VoxelPlane *pl = plane_by_index_[match[i]];
      PointToPlane &pp = ptpl_list_[slot[i]];
      pp.point_b_ = pv.point_b; pp.point_w_ = pv.point_w; pp.normal_ = pl->normal_; pp.center_ = pl->center_; pp.plane_var_ = pl->plane_var_;
      pp.body_cov_ = pv.body_var; pp.layer_ = plane_layer_[match[i]]; pp.d_ = pl->d_; pp.eigen_value_ = 0; pp.is_valid_ = true; pp.dis_to_plane_ = dis[i];
    }
  }
  effct_feat_num_ = (int)ptpl_list_.size();
  if (shim_prof) std::fprintf(stderr, "StateEstimation (full, %d points): set_scan %.3f ms, update + D2H %.3f ms, plane rows %.3f ms (%zu rows), lists %.3f ms (%d threads)\n", n, t_scan, t_update - t_scan, t_planes - t_update,
                              rows.size(), since() - t_planes, threads);
}

void VoxelMapManager::UndistortAndDownsample(const std::vector<PointXYZ> &pcl_wait_proc, const std::vector<float> &curvature, const std::vector<livo2_imu_pose> &IMUpose,
                                             const StatesGroup &state_end, double filter_size_surf) {
  const int n = (int)pcl_wait_proc.size();
  livo2_lidar_cfg cfg{};
  cfg.max_iterations = config_setting_.max_iterations_; cfg.max_layer = config_setting_.max_layer_; cfg.sigma_num = config_setting_.sigma_num_;
  cfg.dept_err = config_setting_.dept_err_; cfg.beam_err = config_setting_.beam_err_; cfg.voxel_size = config_setting_.max_voxel_size_; cfg.deg2rad = 0.0;
  std::memcpy(cfg.extR, extR_.data(), 72); std::memcpy(cfg.extT, extT_.data(), 24);
  feats_down_body_.assign((size_t)std::max(n, 1), PointXYZ{0, 0, 0});
  int32_t n_down = 0;
  dev_.check(livo2_lidar_preprocess_scan(dev_.ctx(), n ? &pcl_wait_proc[0].x : nullptr, curvature.data(), n, IMUpose.data(), (int32_t)IMUpose.size(), state_end.rot_end.data(),
                                         state_end.pos_end.data(), filter_size_surf, &cfg, &n_down, nullptr, &feats_down_body_[0].x));
  feats_down_body_.resize(n_down);
  feats_down_size_ = n_down;
  scan_resident_ = true;
}

void ImuProcess::ForwardPropagate(StatesGroup &state_inout, const std::vector<livo2_imu_step> &steps) {
  livo2_imu_cfg c{};
  std::memcpy(c.cov_gyr, cov_gyr.data(), 24); std::memcpy(c.cov_acc, cov_acc.data(), 24); std::memcpy(c.cov_bias_gyr, cov_bias_gyr.data(), 24); std::memcpy(c.cov_bias_acc, cov_bias_acc.data(), 24);
  c.cov_inv_expo = cov_inv_expo; c.G_m_s2 = 9.81;                                  // reference include/common_lib.h:29
  c.mean_acc_norm = std::sqrt(mean_acc[0] * mean_acc[0] + mean_acc[1] * mean_acc[1] + mean_acc[2] * mean_acc[2]);
  c.ba_bg_est_en = ba_bg_est_en; c.gravity_est_en = gravity_est_en; c.exposure_estimate_en = exposure_estimate_en;
  c.first_call = imu_time_init ? 0 : 1; imu_time_init = true;                      // IMU_Processing.cpp:305-317: tau = 1.0 on the first call
  livo2_state s_in, s_out;
  state_inout.to_abi(s_in);
  std::vector<livo2_imu_pose> pushed(std::max<size_t>(steps.size(), 1));
  dev_.check(livo2_imu_propagate(dev_.ctx(), &s_in, steps.data(), (int32_t)steps.size(), &c, &s_out, pushed.data()));
  state_inout.from_abi(s_out);
  IMUpose.insert(IMUpose.end(), pushed.begin(), pushed.begin() + steps.size());
}

// flat mirror of feat_map on the device: index i <-> VisualPoint* (and, with_obs, observation index <-> Feature*)
void VIOManager::mirrorFeatMap(bool with_obs, const GrayImage *img) {
  if (!feat_map_dirty_ && (!with_obs || obs_resident_)) return;
  std::vector<double> pos; std::vector<int64_t> key; std::vector<uint8_t> act;
  mirror_.clear();
  for (const auto &kv : feat_map)
    for (VisualPoint *pt : kv.second->voxel_points) {
      mirror_.push_back(pt);
      act.push_back(pt != nullptr && !pt->obs_.empty());
      for (int k = 0; k < 3; k++) pos.push_back(pt ? pt->pos_[k] : 0.0);
      key.push_back(kv.first.x); key.push_back(kv.first.y); key.push_back(kv.first.z);
    }
  dev_.check(livo2_visual_map_upload(dev_.ctx(), (int32_t)mirror_.size(), pos.data(), key.data(), act.data()));
  feat_map_dirty_ = false; obs_resident_ = false;
  for (size_t i = 0; i < mirror_.size(); i++) if (mirror_[i]) { mirror_[i]->mirror_index_ = (int32_t)i; mirror_[i]->mirror_dirty_ = false; }
  pending_new_.clear(); pending_new_keys_.clear(); pending_dirty_.clear(); pending_removed_.clear();      // the full upload carries whatever they stood for
  full_syncs_++;
  if (!with_obs) return;
  const size_t n = mirror_.size();
  std::vector<int32_t> off(n + 1, 0), id, iidx, lvl, refp(n, -1);
  std::vector<double> px, f, R, t, ie, nrm(n * 3, 0.0);
  std::vector<float> patch;
  std::vector<uint8_t> ninit(n, 0);
  std::vector<const uint8_t *> imgs;
  obs_mirror_.clear();
  for (size_t i = 0; i < n; i++) {
    const VisualPoint *pt = mirror_[i];
    if (pt) {
      std::memcpy(&nrm[i * 3], pt->normal_.data(), 24);
      ninit[i] = pt->is_normal_initialized_;
      for (Feature *ft : pt->obs_) {
        if (pt->has_ref_patch_ && ft == pt->ref_patch) refp[i] = (int32_t)obs_mirror_.size();
        size_t k = 0; while (k < imgs.size() && imgs[k] != ft->img_) k++;
        if (k == imgs.size()) imgs.push_back(ft->img_);
        ft->mirror_index_ = (int32_t)obs_mirror_.size();
        obs_mirror_.push_back(ft);
        id.push_back(ft->id_); iidx.push_back((int32_t)k); lvl.push_back(ft->level_); ie.push_back(ft->inv_expo_time_);
        px.insert(px.end(), ft->px_.begin(), ft->px_.end()); f.insert(f.end(), ft->f_.begin(), ft->f_.end());
        R.insert(R.end(), ft->R_f_w.begin(), ft->R_f_w.end()); t.insert(t.end(), ft->t_f_w.begin(), ft->t_f_w.end());
        if (!ft->patch_) throw std::runtime_error("retrieveFromVisualSparseMap: Feature without patch_");
        patch.insert(patch.end(), ft->patch_, ft->patch_ + 64);
      }
    }
    off[i + 1] = (int32_t)obs_mirror_.size();
  }
  const size_t bytes = (size_t)img->step * img->rows;
  std::vector<uint8_t> pool(bytes * std::max<size_t>(imgs.size(), 1));
  for (size_t k = 0; k < imgs.size(); k++) std::memcpy(&pool[k * bytes], imgs[k], bytes);
  livo2_visual_obs o{};
  o.n_obs = (int32_t)obs_mirror_.size(); o.n_ref = (int32_t)imgs.size(); o.point_offset = off.data(); o.id = id.data(); o.img_idx = iidx.data();
  o.px = px.data(); o.f = f.data(); o.R = R.data(); o.t = t.data(); o.level = lvl.data(); o.inv_expo = ie.data(); o.patch = patch.data();
  o.normal = nrm.data(); o.normal_initialized = ninit.data(); o.ref_patch = refp.data(); o.ref_imgs = pool.data();
  o.width = img->cols; o.height = img->rows; o.stride = img->step;
  dev_.check(livo2_visual_obs_upload(dev_.ctx(), &o));
  obs_resident_ = true;
  img_slots_ = imgs;
}

// ---- incremental mirror -------------------------------------------------------------------------------------------------------------------------------
void VIOManager::insertPointIntoVoxelMap(VisualPoint *pt_new) {                   // reference src/vio.cpp:227-246
  float loc_xyz[3];
  for (int j = 0; j < 3; j++) {
    loc_xyz[j] = (float)(pt_new->pos_[j] / 0.5);
    if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0f;
  }
  const VOXEL_LOCATION position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
  auto it = feat_map.find(position);
  if (it != feat_map.end()) { it->second->voxel_points.push_back(pt_new); it->second->count++; }
  else { VOXEL_POINTS *ot = new VOXEL_POINTS; ot->voxel_points.push_back(pt_new); feat_map[position] = ot; }
  pt_new->mirror_index_ = -1;
  pending_new_.push_back(pt_new); pending_new_keys_.push_back(position);
}

void VIOManager::markPointDirty(VisualPoint *pt) {
  if (!pt || pt->mirror_dirty_ || pt->mirror_index_ < 0) return;                  // (a point that is not mirrored yet travels whole with the delta that creates it)
  pt->mirror_dirty_ = true;
  pending_dirty_.push_back(pt);
}

void VIOManager::erasePointFromVoxelMap(VisualPoint *pt) {
  if (!pt) return;
  for (auto &kv : feat_map) {                                                       // (the voxel is known to a caller of the reference; the demo's removals are rare)
    auto &v = kv.second->voxel_points;
    auto it = std::find(v.begin(), v.end(), pt);
    if (it != v.end()) { v.erase(it); kv.second->count--; break; }
  }
  if (pt->mirror_index_ >= 0) { mirror_[pt->mirror_index_] = nullptr; pending_removed_.push_back(pt->mirror_index_); }
  for (size_t i = 0; i < pending_new_.size(); i++) if (pending_new_[i] == pt) { pending_new_.erase(pending_new_.begin() + i); pending_new_keys_.erase(pending_new_keys_.begin() + i); break; }
  for (size_t i = 0; i < pending_dirty_.size(); i++) if (pending_dirty_[i] == pt) { pending_dirty_.erase(pending_dirty_.begin() + i); break; }
  pt->mirror_index_ = -1; pt->mirror_dirty_ = false;
}

void VIOManager::syncFeatMap(const GrayImage &img) {
  if (feat_map_dirty_ || !obs_resident_) { mirrorFeatMap(true, &img); return; }
  if (!pending_new_.empty() || !pending_dirty_.empty() || !pending_removed_.empty()) applyPendingDelta(img);
}

void VIOManager::applyPendingDelta(const GrayImage &img) {
  // A delta the device refuses (e.g. LIVO2_ERR_RANGE: an obs_ list longer than the resident stride) leaves the host mirror ahead of the device — the indices assigned
  // below are not undone.  The answer the header prescribes is a full upload: it re-assigns every index, so the failure is turned into one (advisor, round 5).
  auto apply = [&](const livo2_visual_map_delta *dd) {
    if (livo2_visual_map_apply(dev_.ctx(), dd) == LIVO2_OK) return true;
    pending_new_.clear(); pending_new_keys_.clear(); pending_dirty_.clear(); pending_removed_.clear();
    feat_map_dirty_ = true;
    mirrorFeatMap(true, &img);                                                      // throws if even that fails
    return false;
  };
  const size_t n_new = pending_new_.size();
  std::vector<double> npos(n_new * 3), tnormal;
  std::vector<int64_t> nkey(n_new * 3);
  std::vector<uint8_t> nact(n_new, 1), tninit, tactive;
  std::vector<int32_t> oid, oimg, olvl, tpoint, toff(1, 0), tobs, tref;
  std::vector<double> opx, of, oR, ot, oie;
  std::vector<float> opatch;
  std::vector<const uint8_t *> new_imgs;
  for (size_t k = 0; k < n_new; k++) {                                              // new points take the indices behind the resident ones, in queue order
    VisualPoint *pt = pending_new_[k];
    pt->mirror_index_ = (int32_t)mirror_.size(); mirror_.push_back(pt);
    std::memcpy(&npos[k * 3], pt->pos_.data(), 24);
    nkey[k * 3] = pending_new_keys_[k].x; nkey[k * 3 + 1] = pending_new_keys_[k].y; nkey[k * 3 + 2] = pending_new_keys_[k].z;
    pt->mirror_dirty_ = true;
  }
  std::vector<VisualPoint *> touched(pending_new_);                                 // a new point's list / normal / flags travel as a touched row like any other
  touched.insert(touched.end(), pending_dirty_.begin(), pending_dirty_.end());
  auto image_slot = [&](const uint8_t *p) -> int32_t {
    for (size_t k = img_slots_.size(); k-- > 0;) if (img_slots_[k] == p) return (int32_t)k;      // new images sit at the back
    img_slots_.push_back(p); new_imgs.push_back(p);
    return (int32_t)img_slots_.size() - 1;
  };
  for (VisualPoint *pt : touched) {
    tpoint.push_back(pt->mirror_index_);
    int32_t ref = -1;
    for (Feature *ft : pt->obs_) {
      if (ft->mirror_index_ < 0) {                                                  // a Feature the device has not seen: appended behind the resident observations
        if (!ft->patch_) throw std::runtime_error("syncFeatMap: Feature without patch_");
        ft->mirror_index_ = (int32_t)obs_mirror_.size(); obs_mirror_.push_back(ft);
        oid.push_back(ft->id_); oimg.push_back(image_slot(ft->img_)); olvl.push_back(ft->level_); oie.push_back(ft->inv_expo_time_);
        opx.insert(opx.end(), ft->px_.begin(), ft->px_.end()); of.insert(of.end(), ft->f_.begin(), ft->f_.end());
        oR.insert(oR.end(), ft->R_f_w.begin(), ft->R_f_w.end()); ot.insert(ot.end(), ft->t_f_w.begin(), ft->t_f_w.end());
        opatch.insert(opatch.end(), ft->patch_, ft->patch_ + 64);
      }
      tobs.push_back(ft->mirror_index_);
      if (pt->has_ref_patch_ && ft == pt->ref_patch) ref = ft->mirror_index_;
    }
    toff.push_back((int32_t)tobs.size());
    tref.push_back(ref);
    tnormal.insert(tnormal.end(), pt->normal_.begin(), pt->normal_.end());
    tninit.push_back(pt->is_normal_initialized_ ? 1 : 0); tactive.push_back(pt->obs_.empty() ? 0 : 1);
    pt->mirror_dirty_ = false;
  }
  for (int32_t idx : pending_removed_) {                                            // gone from feat_map: the slot stays, inactive, with an empty list
    tpoint.push_back(idx); toff.push_back((int32_t)tobs.size()); tref.push_back(-1);
    tnormal.insert(tnormal.end(), {0.0, 0.0, 0.0}); tninit.push_back(0); tactive.push_back(0);
  }
  const int32_t slot0 = (int32_t)(img_slots_.size() - new_imgs.size());
  for (size_t k = 0; k + 1 < new_imgs.size(); k++) {                                // the API takes one image per call: all but the last travel on their own
    livo2_visual_map_delta di{};
    di.img = new_imgs[k]; di.img_slot = slot0 + (int32_t)k;
    if (!apply(&di)) return;
  }
  livo2_visual_map_delta d{};
  d.n_new_points = (int32_t)n_new; d.n_new_obs = (int32_t)oid.size(); d.n_touched = (int32_t)tpoint.size();
  d.new_pos = npos.data(); d.new_voxel_key = nkey.data(); d.new_active = nact.data();
  d.obs_id = oid.data(); d.obs_img_idx = oimg.data(); d.obs_px = opx.data(); d.obs_f = of.data(); d.obs_R = oR.data(); d.obs_t = ot.data(); d.obs_level = olvl.data();
  d.obs_inv_expo = oie.data(); d.obs_patch = opatch.data();
  d.touched_point = tpoint.data(); d.touched_offset = toff.data(); d.touched_obs = tobs.data(); d.touched_normal = tnormal.data();
  d.touched_normal_initialized = tninit.data(); d.touched_active = tactive.data(); d.touched_ref_patch = tref.data();
  if (!new_imgs.empty()) { d.img = new_imgs.back(); d.img_slot = (int32_t)img_slots_.size() - 1; }
  if (!apply(&d)) return;
  pending_new_.clear(); pending_new_keys_.clear(); pending_dirty_.clear(); pending_removed_.clear();
  delta_syncs_++;
}

void VIOManager::gridSetup() {                                                    // reference src/vio.cpp:67-78
  if (grid_n_width != 0) return;
  if (grid_size > 10) { grid_n_width = (int)std::ceil((double)(width / grid_size)); grid_n_height = (int)std::ceil((double)(height / grid_size)); }
  else { grid_size = height / grid_n_height; grid_n_height = (int)std::ceil((double)(height / grid_size)); grid_n_width = (int)std::ceil((double)(width / grid_size)); }
}

livo2_select_cfg VIOManager::selectCfg() const {
  livo2_select_cfg sc{};
  sc.cam.fx = fx; sc.cam.fy = fy; sc.cam.cx = cx; sc.cam.cy = cy; sc.cam.distortion = cam_distortion_model(); std::memcpy(sc.cam.d, cam_d, sizeof(cam_d)); sc.cam.width = width; sc.cam.height = height;
  std::memcpy(sc.R_cur, R_f_w_new.data(), 72); std::memcpy(sc.t_cur, t_f_w_new.data(), 24);
  sc.border = border; sc.grid_size = grid_size; sc.grid_n_width = grid_n_width; sc.grid_n_height = grid_n_height; sc.patch_size_half = patch_size / 2; sc.raycast_en = raycast_en ? 1 : 0;
  return sc;
}

std::vector<VisualPoint *> VIOManager::selectFromVisualSparseMap(const std::vector<pointWithVar> &pg) {
  if (feat_map.empty()) return {};                                                // reference src/vio.cpp:354
  mirrorFeatMap(false, nullptr);
  gridSetup();
  const int length = grid_n_width * grid_n_height;
  std::vector<double> pgw(pg.size() * 3);
  for (size_t i = 0; i < pg.size(); i++) std::memcpy(&pgw[i * 3], pg[i].point_w.data(), 24);
  const livo2_select_cfg sc = selectCfg();
  std::vector<int32_t> cell(length); std::vector<uint8_t> disc(length);
  map_dist.assign(length, 10000.0f);
  dev_.check(livo2_visual_select(dev_.ctx(), pgw.data(), (int32_t)pg.size(), &sc, cell.data(), map_dist.data(), disc.data(), nullptr));
  std::vector<VisualPoint *> kept;
  for (int i = 0; i < length; i++)
    if (cell[i] >= 0 && !disc[i]) { VisualPoint *pt = mirror_[cell[i]]; if (pt->is_normal_initialized_) kept.push_back(pt); }   // vio.cpp:598-644
  return kept;
}

void VIOManager::updateFrameState(const StatesGroup &s) {                         // reference src/vio.cpp:1690-1697 (+ initializeVIO 57-58 for Rci / Pci)
  M3D Rci; V3D Pci;
  for (int r = 0; r < 3; r++) {                                                   // Rli = extR^T, Pli = -extR^T extT; Rci = Rcl Rli; Pci = Rcl Pli + Pcl
    for (int c = 0; c < 3; c++) Rci[r * 3 + c] = Rcl[r * 3] * extR[c * 3] + Rcl[r * 3 + 1] * extR[c * 3 + 1] + Rcl[r * 3 + 2] * extR[c * 3 + 2];
  }
  V3D Pli;
  for (int r = 0; r < 3; r++) Pli[r] = -(extR[r] * extT[0] + extR[3 + r] * extT[1] + extR[6 + r] * extT[2]);
  for (int r = 0; r < 3; r++) Pci[r] = (Rcl[r * 3] * Pli[0] + Rcl[r * 3 + 1] * Pli[1] + Rcl[r * 3 + 2] * Pli[2]) + Pcl[r];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Rcw[r * 3 + c] = Rci[r * 3] * s.rot_end[c * 3] + Rci[r * 3 + 1] * s.rot_end[c * 3 + 1] + Rci[r * 3 + 2] * s.rot_end[c * 3 + 2];   // Rci * Rwi^T
    Pcw[r] = -(Rcw[r * 3] * s.pos_end[0] + Rcw[r * 3 + 1] * s.pos_end[1] + Rcw[r * 3 + 2] * s.pos_end[2]) + Pci[r];
  }
  R_f_w_new = Rcw; t_f_w_new = Pcw;
}

// retrieveFromVisualSparseMap in three stages (the members r_* carry the chain's outputs between them), so that retrieveAndUpdate below can put the update on the
// GPU before the host lists are built:
//   retrieveChain  — everything up to the device chain's results (total_points known, the survivors resident as the frame of the next update)
//   retrieveLists  — visual_submap's lists and the ref_patch write-backs from those results (host only: ~50 ns per candidate of pointer chasing)
//   retrieveRaycast — add_from_voxel_map (raycast_en; synchronises the stream)
bool VIOManager::retrieveChain(const GrayImage &img, const std::vector<pointWithVar> &pg) {
  if (feat_map.empty()) return false;                                             // reference src/vio.cpp:354
  SubSparseMap &sm = *visual_submap;                                              // visual_submap->reset(), vio.cpp:359
  sm.voxel_points.clear(); sm.search_levels.clear(); sm.errors.clear(); sm.inv_expo_list.clear(); sm.warp_patch.clear();
  static const bool shim_prof = std::getenv("LIVO2_SHIM_PROF") != nullptr;
  const auto tp0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); };
  syncFeatMap(img);
  gridSetup();
  const int length = grid_n_width * grid_n_height, L = patch_pyrimid_level;
  std::vector<double> pgw;
  if (!pg_from_map_update_) {            // (one 24-B field out of every ~900-B pointWithVar: a cache miss per point — spread over a few threads when the scan is large)
    pgw.resize(pg.size() * 3);
    const long long np = (long long)pg.size();
    const int threads = (int)std::max<long long>(1, std::min<long long>(8, np / 4096 + 1));
#pragma omp parallel for schedule(static) num_threads(threads)
    for (long long i = 0; i < np; i++) std::memcpy(&pgw[(size_t)i * 3], pg[(size_t)i].point_w.data(), 24);
  }
  const livo2_select_cfg sc = selectCfg();
  livo2_retrieve_cfg rc{};
  rc.cam = sc.cam;
  std::memcpy(rc.R_cur, R_f_w_new.data(), 72); std::memcpy(rc.t_cur, t_f_w_new.data(), 24);
  rc.inv_expo_cur = state->inv_expo_time; rc.patch_pyrimid_level = L; rc.normal_en = normal_en; rc.ncc_en = ncc_en; rc.ncc_thre = ncc_thre; rc.outlier_threshold = outlier_threshold;
  r_cell_.resize(length); r_cobs_.resize(length); r_acc_.resize(length); r_sl_.resize(length); r_cand_cell_.resize(length); r_err_.resize(length);
  map_dist.assign(length, 10000.0f);
  livo2_retrieve_chain_out out{};
  out.cell_point = r_cell_.data(); out.cell_dist = map_dist.data(); out.cell_obs = r_cobs_.data(); out.cand_cell = r_cand_cell_.data();
  out.tail.accepted = r_acc_.data(); out.tail.search_level = r_sl_.data(); out.tail.error = r_err_.data();
  int32_t n_acc = 0;
  r_n_cand_ = 0;
  const double t_before = since();
  if (pg_from_map_update_) dev_.check(livo2_visual_retrieve_from_map(dev_.ctx(), img.data, img.cols, img.rows, img.step, nullptr, LIVO2_PG_FROM_MAP_UPDATE, &sc, &rc, &out, &r_n_cand_, &n_acc));
  else dev_.check(livo2_visual_retrieve_from_map(dev_.ctx(), img.data, img.cols, img.rows, img.step, pgw.data(), (int32_t)pg.size(), &sc, &rc, &out, &r_n_cand_, &n_acc));
  total_points = n_acc;
  frame_resident_ = true;
  if (shim_prof) std::fprintf(stderr, "retrieveFromVisualSparseMap: before the call %.3f ms, call %.3f ms (chain kernels %.1f us), %d candidates\n", t_before, since() - t_before,
                              livo2_visual_retrieve_from_map_last_kernel_us(dev_.ctx()), r_n_cand_);
  return true;
}

void VIOManager::retrieveLists() {
  SubSparseMap &sm = *visual_submap;
  const size_t na = (size_t)std::max(total_points, 0);
  sm.voxel_points.reserve(na); sm.search_levels.reserve(na); sm.errors.reserve(na); sm.inv_expo_list.reserve(na);
  for (int i = 0; i < r_n_cand_; i++) {
    const int c = r_cand_cell_[i];
    VisualPoint *pt = mirror_[r_cell_[c]]; Feature *ref_ftr = obs_mirror_[r_cobs_[c]];
    if (normal_en) { pt->ref_patch = ref_ftr; pt->has_ref_patch_ = true; }        // vio.cpp:660-661, 689-690
    if (!r_acc_[i]) continue;
    sm.voxel_points.push_back(pt); sm.search_levels.push_back(r_sl_[i]); sm.errors.push_back(r_err_[i]); sm.inv_expo_list.push_back(ref_ftr->inv_expo_time_);   // vio.cpp:762-767
  }
}

void VIOManager::retrieveRaycast() {
  SubSparseMap &sm = *visual_submap;
  sm.add_from_voxel_map.clear();
  if (!raycast_en) return;
  const int length = grid_n_width * grid_n_height;
  std::vector<double> add((size_t)length * 6);
  int32_t n_add = 0;
  dev_.check(livo2_visual_raycast_fetch(dev_.ctx(), add.data(), length, &n_add));
  sm.add_from_voxel_map.resize(n_add);
  for (int k = 0; k < n_add; k++) { std::memcpy(sm.add_from_voxel_map[k].point_w.data(), &add[(size_t)k * 6], 24); std::memcpy(sm.add_from_voxel_map[k].normal.data(), &add[(size_t)k * 6 + 3], 24); }
}

void VIOManager::retrieveFromVisualSparseMap(const GrayImage &img, const std::vector<pointWithVar> &pg) {
  if (!retrieveChain(img, pg)) return;
  retrieveLists();
  retrieveRaycast();
}

// The two calls processFrame makes back to back (reference src/vio.cpp:1808, 1810), with the same results in the same members, in the order that suits a GPU: the
// update is ENQUEUED as soon as the chain has told the host how many patches survived, the host lists are built while it runs, then its result is fetched.
// (computeJacobianAndUpdateEKF reads nothing retrieveLists writes: positions, patches, levels and exposure times of the survivors are resident since the chain.)
void VIOManager::retrieveAndUpdate(const GrayImage &img, const std::vector<pointWithVar> &pg) {
  if (kernel_times_en || inverse_composition_en) { retrieveFromVisualSparseMap(img, pg); computeJacobianAndUpdateEKF(img); return; }   // (both read visual_submap's lists / toggle options around the call)
  if (!retrieveChain(img, pg)) { computeJacobianAndUpdateEKF(img); return; }                          // (an empty feat_map leaves total_points alone, vio.cpp:354, 786)
  retrieveRaycast();
  const bool run = total_points > 0;                                               // vio.cpp:786
  if (run) updateEnqueue(img);
  retrieveLists();
  if (run) updateFetch();
}

void VIOManager::warpAndGateCandidates(const GrayImage &img, const std::vector<Candidate> &cands) {
  const int n = (int)cands.size(), L = patch_pyrimid_level;
  std::vector<double> pos((size_t)n * 3), nrm((size_t)n * 3), px((size_t)n * 2), f((size_t)n * 3), R((size_t)n * 9), t((size_t)n * 3), ie(n);
  std::vector<int32_t> idx(n), lvl(n);
  std::vector<const uint8_t *> imgs;
  for (int i = 0; i < n; i++) {
    const VisualPoint *pt = cands[i].pt; const Feature *ft = cands[i].ref_ftr;
    size_t k = 0; while (k < imgs.size() && imgs[k] != ft->img_) k++;
    if (k == imgs.size()) imgs.push_back(ft->img_);
    idx[i] = (int32_t)k; lvl[i] = ft->level_; ie[i] = ft->inv_expo_time_;
    std::memcpy(&pos[(size_t)i * 3], pt->pos_.data(), 24); std::memcpy(&nrm[(size_t)i * 3], pt->normal_.data(), 24);
    std::memcpy(&px[(size_t)i * 2], ft->px_.data(), 16); std::memcpy(&f[(size_t)i * 3], ft->f_.data(), 24);
    std::memcpy(&R[(size_t)i * 9], ft->R_f_w.data(), 72); std::memcpy(&t[(size_t)i * 3], ft->t_f_w.data(), 24);
  }
  const size_t bytes = (size_t)img.step * img.rows;
  std::vector<uint8_t> pool(bytes * std::max<size_t>(imgs.size(), 1));
  for (size_t k = 0; k < imgs.size(); k++) std::memcpy(&pool[k * bytes], imgs[k], bytes);
  livo2_retrieve_candidates cd{n, 0, pos.data(), nrm.data(), idx.data(), px.data(), f.data(), R.data(), t.data(), lvl.data(), ie.data()};
  livo2_retrieve_cfg rc{};
  rc.cam.fx = fx; rc.cam.fy = fy; rc.cam.cx = cx; rc.cam.cy = cy; rc.cam.distortion = cam_distortion_model(); std::memcpy(rc.cam.d, cam_d, sizeof(cam_d)); rc.cam.width = width; rc.cam.height = height;
  std::memcpy(rc.R_cur, R_f_w_new.data(), 72); std::memcpy(rc.t_cur, t_f_w_new.data(), 24);
  rc.inv_expo_cur = state->inv_expo_time; rc.patch_pyrimid_level = L; rc.normal_en = normal_en; rc.ncc_en = ncc_en; rc.ncc_thre = ncc_thre; rc.outlier_threshold = outlier_threshold;
  std::vector<int32_t> acc(n), sl(n); std::vector<float> err(n);
  livo2_retrieve_out ro{acc.data(), sl.data(), err.data(), nullptr, nullptr, nullptr};
  int32_t n_acc = 0;
  dev_.check(livo2_visual_retrieve_warp(dev_.ctx(), img.data, img.cols, img.rows, img.step, pool.data(), (int32_t)imgs.size(), &cd, &rc, &ro, &n_acc));
  SubSparseMap &sm = *visual_submap;       // reference src/vio.cpp:762-767 (warp_patch stays on the device)
  sm.voxel_points.clear(); sm.search_levels.clear(); sm.errors.clear(); sm.inv_expo_list.clear(); sm.warp_patch.clear();
  for (int i = 0; i < n; i++) if (acc[i]) {
    sm.voxel_points.push_back(cands[i].pt); sm.search_levels.push_back(sl[i]); sm.errors.push_back(err[i]); sm.inv_expo_list.push_back(cands[i].ref_ftr->inv_expo_time_);
  }
  total_points = n_acc;
  frame_resident_ = true;
}

void VIOManager::computeJacobianAndUpdateEKF(const GrayImage &img) {
  if (total_points == 0) return;            // reference src/vio.cpp:786
  updateEnqueue(img);
  updateFetch();
}

// computeJacobianAndUpdateEKF up to the enqueue of the update (livo2_visual_update_async) ...
void VIOManager::updateEnqueue(const GrayImage &img) {
  const int M = total_points, L = patch_pyrimid_level;
  u_t0_ = std::chrono::steady_clock::now();
  if (!frame_resident_) {                   // (a sub-map the retrieval left on the device needs none of this: the 1 MB of warp_patch alone was ~0.05 ms of zero-fill per frame)
    std::vector<double> pos((size_t)M * 3);
    std::vector<float> warp((size_t)M * L * 64);
    for (int i = 0; i < M; i++) {
      std::memcpy(&pos[(size_t)i * 3], visual_submap->voxel_points[i]->pos_.data(), 24);
      std::memcpy(&warp[(size_t)i * L * 64], visual_submap->warp_patch[i].data(), (size_t)L * 64 * sizeof(float));   // ragged vector<vector<float>> -> [M][L][64]
    }
    dev_.check(livo2_visual_set_frame(dev_.ctx(), img.data, img.cols, img.rows, img.step, pos.data(), warp.data(), visual_submap->search_levels.data(),
                                      visual_submap->inv_expo_list.data(), M, L));
  }
  frame_resident_ = false;
  if (inverse_composition_en) {             // gather the reference patches (distinct reference images are uploaded once each)
    std::vector<const uint8_t *> imgs; std::vector<int32_t> idx(M);
    std::vector<double> px((size_t)M * 2), f((size_t)M * 3), R((size_t)M * 9), rp((size_t)M * 3);
    for (int i = 0; i < M; i++) {
      const Feature *ft = visual_submap->voxel_points[i]->ref_patch;
      if (!ft || !ft->img_) throw std::runtime_error("inverse_composition_en: VisualPoint without ref_patch");
      size_t k = 0; while (k < imgs.size() && imgs[k] != ft->img_) k++;
      if (k == imgs.size()) imgs.push_back(ft->img_);
      idx[i] = (int32_t)k;
      std::memcpy(&px[(size_t)i * 2], ft->px_.data(), 16); std::memcpy(&f[(size_t)i * 3], ft->f_.data(), 24);
      std::memcpy(&R[(size_t)i * 9], ft->R_f_w.data(), 72); std::memcpy(&rp[(size_t)i * 3], ft->pos.data(), 24);
    }
    const size_t bytes = (size_t)img.step * img.rows;
    std::vector<uint8_t> stack(bytes * imgs.size());
    for (size_t k = 0; k < imgs.size(); k++) std::memcpy(&stack[k * bytes], imgs[k], bytes);
    dev_.check(livo2_visual_set_reference(dev_.ctx(), stack.data(), (int32_t)imgs.size(), idx.data(), px.data(), f.data(), R.data(), rp.data()));
  }
  livo2_visual_cfg cfg{};
  cfg.cam.fx = fx; cfg.cam.fy = fy; cfg.cam.cx = cx; cfg.cam.cy = cy; cfg.cam.distortion = cam_distortion_model(); std::memcpy(cfg.cam.d, cam_d, sizeof(cam_d)); cfg.cam.width = width; cfg.cam.height = height;
  std::memcpy(cfg.Rcl, Rcl.data(), 72); std::memcpy(cfg.Pcl, Pcl.data(), 24); std::memcpy(cfg.extR, extR.data(), 72); std::memcpy(cfg.extT, extT.data(), 24);
  cfg.img_point_cov = img_point_cov; cfg.patch_pyrimid_level = L; cfg.max_iterations = max_iterations;
  cfg.exposure_estimate_en = exposure_estimate_en; cfg.inverse_composition_en = inverse_composition_en; cfg.mp_proc_num = mp_proc_num;
  compute_jacobian_time = update_ekf_time = 0.0;                                            // vio.cpp:788
  // the residual / solve split only exists in the launch-per-step sequence: with the times requested that sequence runs instead of the one-launch update
  if (kernel_times_en) { dev_.check(livo2_ctx_set_option(dev_.ctx(), "visual_persistent", 0)); dev_.check(livo2_ctx_kernel_timing(dev_.ctx(), 1)); livo2_ctx_kernel_timing_read(dev_.ctx(), 1, nullptr, nullptr, 1); livo2_ctx_kernel_timing_read(dev_.ctx(), 3, nullptr, nullptr, 1); }
  livo2_state s_in, s_prop;
  state->to_abi(s_in); state_propagat->to_abi(s_prop);
  dev_.check(livo2_visual_update_async(dev_.ctx(), &s_in, &s_prop, &cfg));
  update_enqueued_at_ = std::chrono::steady_clock::now();
}

// ... and from its fetch on
void VIOManager::updateFetch() {
  static const bool shim_prof = std::getenv("LIVO2_SHIM_PROF") != nullptr;
  const int M = total_points;
  static livo2_visual_result res;
  visual_submap->errors.resize(M);
  dev_.check(livo2_visual_update_fetch(dev_.ctx(), &res, visual_submap->errors.data()));
  if (shim_prof) std::fprintf(stderr, "computeJacobianAndUpdateEKF: %.3f ms from the enqueue to the fetched result (%d steps)\n",
                              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - u_t0_).count(), res.n_steps);
  if (kernel_times_en) {
    double ms = 0; int64_t nl = 0;
    dev_.check(livo2_ctx_kernel_timing_read(dev_.ctx(), 1, &ms, &nl, 1)); compute_jacobian_time = ms * 1e-3;
    dev_.check(livo2_ctx_kernel_timing_read(dev_.ctx(), 3, &ms, &nl, 1)); update_ekf_time = ms * 1e-3;
    dev_.check(livo2_ctx_kernel_timing(dev_.ctx(), 0));
    dev_.check(livo2_ctx_set_option(dev_.ctx(), "visual_persistent", 1));
  }
  state->from_abi(res.state);
  std::memcpy(G.data(), res.G, sizeof(res.G));
  std::memcpy(Rcw.data(), res.Rcw, 72); std::memcpy(Pcw.data(), res.Pcw, 24);     // new_frame_->T_f_w_ = SE3(Rcw, Pcw)
  H_T_H.fill(0.0);
  for (int k = res.n_steps - 1; k >= 0; k--)
    if (res.steps[k].accepted) { for (int r = 0; r < 7; r++) for (int c = 0; c < 7; c++) H_T_H[r * LIVO2_DIM_STATE + c] = res.steps[k].HtH[r * 7 + c]; break; }
}

} // namespace livo2
