// See livo2_host.hpp.  Pure data movement between the reference-shaped containers and the C ABI.
#include "livo2_host.hpp"

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
  if (map_dirty_) FlattenAndUpload();
  const int n = (int)feats_down_body_.size();
  feats_down_size_ = n;
  livo2_lidar_cfg cfg{};
  cfg.max_iterations = config_setting_.max_iterations_; cfg.max_layer = config_setting_.max_layer_; cfg.sigma_num = config_setting_.sigma_num_;
  cfg.dept_err = config_setting_.dept_err_; cfg.beam_err = config_setting_.beam_err_; cfg.voxel_size = config_setting_.max_voxel_size_; cfg.deg2rad = 0.0;
  std::memcpy(cfg.extR, extR_.data(), 72); std::memcpy(cfg.extT, extT_.data(), 24);
  static_assert(sizeof(PointXYZ) == 12, "xyz AoS");
  dev_.check(livo2_lidar_set_scan(dev_.ctx(), n ? &feats_down_body_[0].x : nullptr, n, &cfg));

  std::vector<int32_t> match(n), normal_plane(n);
  std::vector<float> dis(n), pw((size_t)n * 3);
  std::vector<double> var((size_t)n * 9), bcov((size_t)n * 9);
  livo2_lidar_points pts{};
  pts.match_plane = match.data(); pts.dis_to_plane = dis.data(); pts.point_w = pw.data(); pts.normal_plane = normal_plane.data();
  pts.var = var.data(); pts.body_cov = bcov.data();
  livo2_state s_in, s_prop;
  state_.to_abi(s_in); state_propagat.to_abi(s_prop);
  livo2_lidar_result res;
  dev_.check(livo2_lidar_update(dev_.ctx(), &s_in, &s_prop, &cfg, &res, &pts));
  state_.from_abi(res.state);
  std::memcpy(position_last_.data(), res.position_last, 24);

  // what the reference leaves behind for LIVMapper (src/LIVMapper.cpp:371-426, 446) and VIO (src/vio.cpp:811)
  pv_list_.assign(n, pointWithVar());
  cross_mat_list_.resize(n); body_cov_list_.resize(n);
  ptpl_list_.clear();
  for (int i = 0; i < n; i++) {
    pointWithVar &pv = pv_list_[i];
    const PointXYZ &p = feats_down_body_[i];
    pv.point_b = {p.x, p.y, p.z};
    pv.point_w = {pw[(size_t)i * 3], pw[(size_t)i * 3 + 1], pw[(size_t)i * 3 + 2]};
    std::memcpy(pv.var.data(), &var[(size_t)i * 9], 72);
    std::memcpy(pv.body_var.data(), &bcov[(size_t)i * 9], 72);
    body_cov_list_[i] = pv.body_var;
    double pz = (p.z == 0) ? 0.001 : (double)p.z;                                   // reference src/voxel_map.cpp:352-358
    double q[3];
    for (int j = 0; j < 3; j++) q[j] = extR_[j * 3] * p.x + extR_[j * 3 + 1] * p.y + extR_[j * 3 + 2] * pz + extT_[j];
    cross_mat_list_[i] = {0.0, -q[2], q[1], q[2], 0.0, -q[0], -q[1], q[0], 0.0};
    if (normal_plane[i] >= 0) pv.normal = plane_by_index_[normal_plane[i]]->normal_;
    if (match[i] >= 0) {
      const VoxelPlane *pl = plane_by_index_[match[i]];
      PointToPlane pp;
      pp.point_b_ = pv.point_b; pp.point_w_ = pv.point_w; pp.normal_ = pl->normal_; pp.center_ = pl->center_; pp.plane_var_ = pl->plane_var_;
      pp.body_cov_ = pv.body_var; pp.layer_ = plane_layer_[match[i]]; pp.d_ = pl->d_; pp.is_valid_ = true; pp.dis_to_plane_ = dis[i];
      ptpl_list_.push_back(pp);
    }
  }
  effct_feat_num_ = (int)ptpl_list_.size();
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
  rc.cam.fx = fx; rc.cam.fy = fy; rc.cam.cx = cx; rc.cam.cy = cy; rc.cam.distortion = 0; rc.cam.width = width; rc.cam.height = height;
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
  const int M = total_points, L = patch_pyrimid_level;
  std::vector<double> pos((size_t)M * 3);
  std::vector<float> warp((size_t)M * L * 64);
  for (int i = 0; i < M && !frame_resident_; i++) {
    std::memcpy(&pos[(size_t)i * 3], visual_submap->voxel_points[i]->pos_.data(), 24);
    std::memcpy(&warp[(size_t)i * L * 64], visual_submap->warp_patch[i].data(), (size_t)L * 64 * sizeof(float));   // ragged vector<vector<float>> -> [M][L][64]
  }
  if (!frame_resident_)
    dev_.check(livo2_visual_set_frame(dev_.ctx(), img.data, img.cols, img.rows, img.step, pos.data(), warp.data(), visual_submap->search_levels.data(),
                                      visual_submap->inv_expo_list.data(), M, L));
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
  cfg.cam.fx = fx; cfg.cam.fy = fy; cfg.cam.cx = cx; cfg.cam.cy = cy; cfg.cam.distortion = 0; cfg.cam.width = width; cfg.cam.height = height;
  std::memcpy(cfg.Rcl, Rcl.data(), 72); std::memcpy(cfg.Pcl, Pcl.data(), 24); std::memcpy(cfg.extR, extR.data(), 72); std::memcpy(cfg.extT, extT.data(), 24);
  cfg.img_point_cov = img_point_cov; cfg.patch_pyrimid_level = L; cfg.max_iterations = max_iterations;
  cfg.exposure_estimate_en = exposure_estimate_en; cfg.inverse_composition_en = inverse_composition_en;
  livo2_state s_in, s_prop;
  state->to_abi(s_in); state_propagat->to_abi(s_prop);
  static livo2_visual_result res;
  visual_submap->errors.resize(M);
  dev_.check(livo2_visual_update(dev_.ctx(), &s_in, &s_prop, &cfg, &res, visual_submap->errors.data()));
  state->from_abi(res.state);
  std::memcpy(G.data(), res.G, sizeof(res.G));
  std::memcpy(Rcw.data(), res.Rcw, 72); std::memcpy(Pcw.data(), res.Pcw, 24);     // new_frame_->T_f_w_ = SE3(Rcw, Pcw)
  H_T_H.fill(0.0);
  for (int k = res.n_steps - 1; k >= 0; k--)
    if (res.steps[k].accepted) { for (int r = 0; r < 7; r++) for (int c = 0; c < 7; c++) H_T_H[r * LIVO2_DIM_STATE + c] = res.steps[k].HtH[r * 7 + c]; break; }
}

} // namespace livo2
