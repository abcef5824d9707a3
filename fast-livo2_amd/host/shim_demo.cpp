// Drives the C++ host shim exactly the way LIVMapper drives the reference managers (src/LIVMapper.cpp:357-372, src/vio.cpp:1806-1810):
// fills a VoxelMapManager (pointer-based voxel_map_, feats_down_body_, state_) / a VIOManager (visual_submap, *state) from a dump
// directory written by tests/test_host_shim_gpu.py, calls StateEstimation / computeJacobianAndUpdateEKF, and writes what the rest
// of the pipeline would read back.  Usage: shim_demo <dir>
#include <algorithm>
#include <array>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

#include "livo2_host.hpp"

using namespace livo2;

template <typename T> static std::vector<T> rd(const std::string &dir, const char *name) {
  std::ifstream f(dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
  if (!f) return {};
  size_t bytes = (size_t)f.tellg(); f.seekg(0);
  std::vector<T> v(bytes / sizeof(T));
  f.read((char *)v.data(), bytes);
  return v;
}
template <typename T> static void wr(const std::string &dir, const char *name, const T *p, size_t n) {
  std::ofstream f(dir + "/" + name + ".bin", std::ios::binary);
  f.write((const char *)p, n * sizeof(T));
}
static StatesGroup state_from(const std::vector<double> &v) {       // 25 scalars + 361
  StatesGroup s; livo2_state a;
  std::memcpy(&a, v.data(), sizeof(a)); s.from_abi(a); return s;
}
static std::vector<double> state_to(const StatesGroup &s) { livo2_state a; s.to_abi(a); std::vector<double> v(sizeof(a) / 8); std::memcpy(v.data(), &a, sizeof(a)); return v; }

static std::vector<VoxelOctoTree *> g_node_of_plane;     // plane index of the dump -> its octree node (for the FitPlanes leg)

static VoxelOctoTree *build(int node, int layer, const std::vector<int32_t> &node_plane, const std::vector<int32_t> &node_child, const std::vector<double> &pn,
                            const std::vector<double> &pc, const std::vector<double> &pv, const std::vector<float> &pd, const std::vector<float> &pr) {
  VoxelOctoTree *t = new VoxelOctoTree; t->layer_ = layer;
  int pi = node_plane[node];
  if (pi >= 0) {
    VoxelPlane *p = t->plane_ptr_;
    for (int k = 0; k < 3; k++) { p->normal_[k] = pn[pi * 3 + k]; p->center_[k] = pc[pi * 3 + k]; }
    for (int k = 0; k < 36; k++) p->plane_var_[k] = pv[(size_t)pi * 36 + k];
    p->d_ = pd[pi]; p->radius_ = pr[pi]; p->is_plane_ = true;
    if ((size_t)pi >= g_node_of_plane.size()) g_node_of_plane.resize(pi + 1, nullptr);
    g_node_of_plane[pi] = t;
  }
  for (int k = 0; k < 8; k++) { int c = node_child[(size_t)node * 8 + k]; if (c >= 0) t->leaves_[k] = build(c, layer + 1, node_plane, node_child, pn, pc, pv, pd, pr); }
  return t;
}

// flat dump of a VoxelMapManager's map in the layout of livo2_map_view (roots sorted by key, nodes depth-first)
struct FlatOut { std::vector<int64_t> root_key; std::vector<int32_t> root_node, node_plane, node_child; std::vector<double> root_center, pn, pc, pv; std::vector<float> root_quarter, pd, pr; };
static int32_t dump_node(const VoxelOctoTree *t, FlatOut &f) {
  const int32_t me = (int32_t)f.node_plane.size();
  f.node_plane.push_back(-1); f.node_child.insert(f.node_child.end(), 8, -1);
  const VoxelPlane *p = t->plane_ptr_;
  if (p->is_plane_) {
    f.node_plane[me] = (int32_t)f.pd.size();
    f.pn.insert(f.pn.end(), p->normal_.begin(), p->normal_.end()); f.pc.insert(f.pc.end(), p->center_.begin(), p->center_.end());
    f.pv.insert(f.pv.end(), p->plane_var_.begin(), p->plane_var_.end()); f.pd.push_back(p->d_); f.pr.push_back(p->radius_);
  }
  for (int k = 0; k < 8; k++) if (t->leaves_[k]) { const int32_t c = dump_node(t->leaves_[k], f); f.node_child[(size_t)me * 8 + k] = c; }
  return me;
}
static void dump_map(const VoxelMapManager &vm, const std::string &dir, const std::string &prefix) {
  std::vector<std::pair<std::array<int64_t, 3>, const VoxelOctoTree *>> roots;
  for (const auto &kv : vm.voxel_map_) roots.push_back({{kv.first.x, kv.first.y, kv.first.z}, kv.second});
  std::sort(roots.begin(), roots.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
  FlatOut f;
  for (const auto &r : roots) {
    f.root_key.insert(f.root_key.end(), r.first.begin(), r.first.end());
    for (int k = 0; k < 3; k++) f.root_center.push_back(r.second->voxel_center_[k]);
    f.root_quarter.push_back(r.second->quater_length_);
    f.root_node.push_back(dump_node(r.second, f));
  }
  wr(dir, (prefix + "root_key").c_str(), f.root_key.data(), f.root_key.size()); wr(dir, (prefix + "root_node").c_str(), f.root_node.data(), f.root_node.size());
  wr(dir, (prefix + "root_center").c_str(), f.root_center.data(), f.root_center.size()); wr(dir, (prefix + "root_quarter").c_str(), f.root_quarter.data(), f.root_quarter.size());
  wr(dir, (prefix + "node_plane").c_str(), f.node_plane.data(), f.node_plane.size()); wr(dir, (prefix + "node_child").c_str(), f.node_child.data(), f.node_child.size());
  wr(dir, (prefix + "plane_normal").c_str(), f.pn.data(), f.pn.size()); wr(dir, (prefix + "plane_center").c_str(), f.pc.data(), f.pc.size());
  wr(dir, (prefix + "plane_var").c_str(), f.pv.data(), f.pv.size()); wr(dir, (prefix + "plane_d").c_str(), f.pd.data(), f.pd.size()); wr(dir, (prefix + "plane_radius").c_str(), f.pr.data(), f.pr.size());
}
// the device-resident map (livo2_map_tree_export) in the same files
static void dump_device_map(Device &dev, const std::string &dir, const std::string &prefix) {
  int32_t c[8];
  dev.check(livo2_map_tree_stats(dev.ctx(), c));
  const size_t R = (size_t)c[7], N = (size_t)std::max(c[0], 1), P = (size_t)std::max(c[2], 1);
  std::vector<int64_t> root_key(R * 3); std::vector<int32_t> root_node(R), node_plane(N), node_child(N * 8);
  std::vector<double> root_center(R * 3), pn(P * 3), pc(P * 3), pv(P * 36); std::vector<float> root_quarter(R), pd(P), pr(P);
  dev.check(livo2_map_tree_export(dev.ctx(), root_key.data(), root_node.data(), root_center.data(), root_quarter.data(), node_plane.data(), node_child.data(), pn.data(), pc.data(),
                                  pv.data(), pd.data(), pr.data(), nullptr));
  wr(dir, (prefix + "root_key").c_str(), root_key.data(), root_key.size()); wr(dir, (prefix + "root_node").c_str(), root_node.data(), root_node.size());
  wr(dir, (prefix + "root_center").c_str(), root_center.data(), root_center.size()); wr(dir, (prefix + "root_quarter").c_str(), root_quarter.data(), root_quarter.size());
  wr(dir, (prefix + "node_plane").c_str(), node_plane.data(), (size_t)c[0]); wr(dir, (prefix + "node_child").c_str(), node_child.data(), (size_t)c[0] * 8);
  wr(dir, (prefix + "plane_normal").c_str(), pn.data(), (size_t)c[2] * 3); wr(dir, (prefix + "plane_center").c_str(), pc.data(), (size_t)c[2] * 3);
  wr(dir, (prefix + "plane_var").c_str(), pv.data(), (size_t)c[2] * 36); wr(dir, (prefix + "plane_d").c_str(), pd.data(), (size_t)c[2]); wr(dir, (prefix + "plane_radius").c_str(), pr.data(), (size_t)c[2]);
}
static std::vector<pointWithVar> points_from(const std::vector<double> &pw, const std::vector<double> &var) {
  std::vector<pointWithVar> v(pw.size() / 3);
  for (size_t i = 0; i < v.size(); i++) { for (int k = 0; k < 3; k++) v[i].point_w[k] = pw[i * 3 + k]; for (int k = 0; k < 9; k++) v[i].var[k] = var[i * 9 + k]; }
  return v;
}

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: shim_demo <dir>\n"); return 2; }
  const std::string dir = argv[1];
  try {
    Device dev(0);
    // ---- LiDAR ------------------------------------------------------------------------------------------------------------
    auto keys = rd<int64_t>(dir, "root_key");
    if (!keys.empty()) {
      auto root_node = rd<int32_t>(dir, "root_node"); auto rc = rd<double>(dir, "root_center"); auto rq = rd<float>(dir, "root_quarter");
      auto node_plane = rd<int32_t>(dir, "node_plane"), node_child = rd<int32_t>(dir, "node_child");
      auto pn = rd<double>(dir, "plane_normal"), pc = rd<double>(dir, "plane_center"), pv = rd<double>(dir, "plane_var");
      auto pd = rd<float>(dir, "plane_d"), pr = rd<float>(dir, "plane_radius");
      auto xyz = rd<float>(dir, "xyz"); auto cfgv = rd<double>(dir, "lidar_cfg");     // max_it, max_layer, sigma, dept, beam, voxel, extR9, extT3
      VoxelMapManager vm(dev);
      for (size_t r = 0; r < root_node.size(); r++) {
        VoxelOctoTree *t = build(root_node[r], 0, node_plane, node_child, pn, pc, pv, pd, pr);
        for (int k = 0; k < 3; k++) t->voxel_center_[k] = rc[r * 3 + k];
        t->quater_length_ = rq[r];
        vm.voxel_map_[VOXEL_LOCATION(keys[r * 3], keys[r * 3 + 1], keys[r * 3 + 2])] = t;
      }
      vm.config_setting_.max_iterations_ = (int)cfgv[0]; vm.config_setting_.max_layer_ = (int)cfgv[1]; vm.config_setting_.sigma_num_ = cfgv[2];
      vm.config_setting_.dept_err_ = cfgv[3]; vm.config_setting_.beam_err_ = cfgv[4]; vm.config_setting_.max_voxel_size_ = cfgv[5];
      for (int k = 0; k < 9; k++) vm.extR_[k] = cfgv[6 + k];
      for (int k = 0; k < 3; k++) vm.extT_[k] = cfgv[15 + k];
      vm.feats_down_body_.resize(xyz.size() / 3);
      std::memcpy(vm.feats_down_body_.data(), xyz.data(), xyz.size() * 4);
      vm.state_ = state_from(rd<double>(dir, "state_in"));
      StatesGroup prop = state_from(rd<double>(dir, "state_prop"));
      vm.StateEstimation(prop);                                        // <- the reference call site (src/LIVMapper.cpp:370)
      auto so = state_to(vm.state_);
      wr(dir, "out_state", so.data(), so.size());
      std::vector<double> normals, pvvar; std::vector<float> dis; std::vector<double> ptw;
      for (auto &pvx : vm.pv_list_) { normals.insert(normals.end(), pvx.normal.begin(), pvx.normal.end()); pvvar.insert(pvvar.end(), pvx.var.begin(), pvx.var.end()); }
      for (auto &pp : vm.ptpl_list_) { dis.push_back(pp.dis_to_plane_); ptw.insert(ptw.end(), pp.point_w_.begin(), pp.point_w_.end()); }
      wr(dir, "out_pv_normal", normals.data(), normals.size()); wr(dir, "out_pv_var", pvvar.data(), pvvar.size());
      wr(dir, "out_ptpl_dis", dis.data(), dis.size()); wr(dir, "out_ptpl_pw", ptw.data(), ptw.size());
      int32_t eff = vm.effct_feat_num_; wr(dir, "out_effct", &eff, 1);
      wr(dir, "out_geoquat", vm.geoQuat_.data(), 4);
      std::printf("lidar: effct_feat_num_=%d\n", vm.effct_feat_num_);
      // ---- UpdateVoxelMap's re-fits in bulk (src/voxel_map.cpp:55-135 via VoxelMapManager::FitPlanes), then the next frame's update ----
      auto fit_plane = rd<int32_t>(dir, "fit_plane");
      if (!fit_plane.empty()) {
        auto fpw = rd<double>(dir, "fit_pw"), fvar = rd<double>(dir, "fit_var"); auto foff = rd<int32_t>(dir, "fit_off");
        std::vector<VoxelOctoTree *> voxels;
        for (size_t g = 0; g < fit_plane.size(); g++) {
          VoxelOctoTree *t = g_node_of_plane[fit_plane[g]];
          t->temp_points_.clear();
          for (int i = foff[g]; i < foff[g + 1]; i++) {
            pointWithVar pv;
            for (int k = 0; k < 3; k++) pv.point_w[k] = fpw[(size_t)i * 3 + k];
            for (int k = 0; k < 9; k++) pv.var[k] = fvar[(size_t)i * 9 + k];
            t->temp_points_.push_back(pv);
          }
          voxels.push_back(t);
        }
        vm.FitPlanes(voxels);
        std::vector<double> fc, fn; std::vector<float> fr; std::vector<int32_t> fis;
        for (auto *t : voxels) { const VoxelPlane *p = t->plane_ptr_; fc.insert(fc.end(), p->center_.begin(), p->center_.end()); fn.insert(fn.end(), p->normal_.begin(), p->normal_.end()); fr.push_back(p->radius_); fis.push_back(p->is_plane_); }
        wr(dir, "fit_out_center", fc.data(), fc.size()); wr(dir, "fit_out_normal", fn.data(), fn.size()); wr(dir, "fit_out_radius", fr.data(), fr.size()); wr(dir, "fit_out_is_plane", fis.data(), fis.size());
        vm.state_ = state_from(rd<double>(dir, "state_in"));
        vm.StateEstimation(prop);
        auto so2 = state_to(vm.state_);
        wr(dir, "out_state2", so2.data(), so2.size());
        std::printf("fit: %zu voxels re-fitted\n", voxels.size());
      }
    }
    // ---- IMU forward propagation, then undistortion + down-sampling of a raw scan with the poses it produced (src/IMU_Processing.cpp:298-539) ----
    auto imu_steps = rd<double>(dir, "imu_steps");     // [n][8]
    if (!imu_steps.empty()) {
      ImuProcess imu(dev);
      auto ic = rd<double>(dir, "imu_cfg");            // cov_gyr3 cov_acc3 cov_bias_gyr3 cov_bias_acc3 cov_inv_expo mean_acc3 flags3
      for (int k = 0; k < 3; k++) { imu.cov_gyr[k] = ic[k]; imu.cov_acc[k] = ic[3 + k]; imu.cov_bias_gyr[k] = ic[6 + k]; imu.cov_bias_acc[k] = ic[9 + k]; imu.mean_acc[k] = ic[13 + k]; }
      imu.cov_inv_expo = ic[12]; imu.ba_bg_est_en = ic[16] != 0; imu.gravity_est_en = ic[17] != 0; imu.exposure_estimate_en = ic[18] != 0;
      StatesGroup st = state_from(rd<double>(dir, "imu_state_in"));
      std::vector<livo2_imu_step> steps(imu_steps.size() / 8);
      std::memcpy(steps.data(), imu_steps.data(), imu_steps.size() * 8);
      livo2_imu_pose first{};                            // IMUpose[0] = set_pose6d(0.0, acc_s_last, angvel_last, vel_end, pos_end, rot_end) (IMU_Processing.cpp:281)
      for (int k = 0; k < 3; k++) { first.vel[k] = st.vel_end[k]; first.pos[k] = st.pos_end[k]; }
      for (int k = 0; k < 9; k++) first.rot[k] = st.rot_end[k];
      imu.IMUpose.push_back(first);
      imu.ForwardPropagate(st, steps);
      auto so = state_to(st);
      wr(dir, "imu_out_state", so.data(), so.size());
      wr(dir, "imu_out_poses", (const double *)imu.IMUpose.data(), imu.IMUpose.size() * 22);
      auto raw = rd<float>(dir, "raw_xyz"); auto cur = rd<float>(dir, "raw_curvature");
      if (!raw.empty()) {
        auto rc = rd<double>(dir, "raw_cfg");            // extR9 extT3 leaf voxel_size
        VoxelMapManager vm(dev);
        for (int k = 0; k < 9; k++) vm.extR_[k] = rc[k];
        for (int k = 0; k < 3; k++) vm.extT_[k] = rc[9 + k];
        vm.config_setting_.max_voxel_size_ = rc[13];
        std::vector<PointXYZ> pts(raw.size() / 3);
        std::memcpy(pts.data(), raw.data(), raw.size() * 4);
        vm.UndistortAndDownsample(pts, cur, imu.IMUpose, st, rc[12]);
        wr(dir, "raw_out_down", &vm.feats_down_body_[0].x, vm.feats_down_body_.size() * 3);
        std::printf("pre-stage: %zu raw points -> %d\n", pts.size(), vm.feats_down_size_);
      }
    }
    // ---- BuildVoxelMap + UpdateVoxelMap with device-side plane fits (src/voxel_map.cpp:532-591, 609-641) ---------------------------
    auto bld_pw = rd<double>(dir, "bld_pw");
    if (!bld_pw.empty()) {
      auto mc = rd<double>(dir, "map_cfg");       // voxel_size, max_layer, max_points_num, planner_threshold, layer_init_num x5
      VoxelMapManager vm(dev);
      vm.config_setting_.max_voxel_size_ = mc[0]; vm.config_setting_.max_layer_ = (int)mc[1]; vm.config_setting_.max_points_num_ = (int)mc[2];
      vm.config_setting_.planner_threshold_ = mc[3];
      vm.config_setting_.layer_init_num_.assign(5, 5);
      for (int k = 0; k < 5; k++) vm.config_setting_.layer_init_num_[k] = (int)mc[4 + k];
      vm.BuildVoxelMap(points_from(bld_pw, rd<double>(dir, "bld_var")));
      std::printf("BuildVoxelMap: %zu roots, %d plane fits in %d device batches\n", vm.voxel_map_.size(), vm.last_fit_count_, vm.last_fit_rounds_);
      dump_map(vm, dir, "bld_out_");
      int32_t st[2] = {vm.last_fit_count_, vm.last_fit_rounds_}; wr(dir, "bld_out_stats", st, 2);
      auto upd_pw = rd<double>(dir, "upd_pw");
      if (!upd_pw.empty()) {
        vm.UpdateVoxelMap(points_from(upd_pw, rd<double>(dir, "upd_var")));
        std::printf("UpdateVoxelMap: %zu roots, %d plane fits in %d device batches\n", vm.voxel_map_.size(), vm.last_fit_count_, vm.last_fit_rounds_);
        dump_map(vm, dir, "upd_out_");
        int32_t su[2] = {vm.last_fit_count_, vm.last_fit_rounds_}; wr(dir, "upd_out_stats", su, 2);
      }
    }
    // ---- a LIO sequence the way LIVMapper::handleLIO drives it (src/LIVMapper.cpp:357-426): per frame StateEstimation on the propagated
    // state, then the world points / covariances of the scan from the posterior (413-423) and UpdateVoxelMap with them ---------------------
    auto seq_counts = rd<int32_t>(dir, "seq_counts");
    if (!seq_counts.empty()) {
      auto mc = rd<double>(dir, "seq_map_cfg"); auto lc = rd<double>(dir, "seq_lidar_cfg");
      auto scans = rd<float>(dir, "seq_scans"); auto motion = rd<double>(dir, "seq_motion"); auto qd = rd<double>(dir, "seq_q");
      VoxelMapManager vm(dev);
      vm.device_map_ = !rd<int32_t>(dir, "seq_device_map").empty();      // the octree on the GPU (livo2_map_tree_*) instead of on the host
      vm.config_setting_.max_voxel_size_ = mc[0]; vm.config_setting_.max_layer_ = (int)mc[1]; vm.config_setting_.max_points_num_ = (int)mc[2]; vm.config_setting_.planner_threshold_ = mc[3];
      vm.config_setting_.layer_init_num_.assign(5, 5);
      for (int k = 0; k < 5; k++) vm.config_setting_.layer_init_num_[k] = (int)mc[4 + k];
      vm.config_setting_.max_iterations_ = (int)lc[0]; vm.config_setting_.sigma_num_ = lc[2]; vm.config_setting_.dept_err_ = lc[3]; vm.config_setting_.beam_err_ = lc[4];
      for (int k = 0; k < 9; k++) vm.extR_[k] = lc[6 + k];
      for (int k = 0; k < 3; k++) vm.extT_[k] = lc[15 + k];
      auto slide = rd<double>(dir, "seq_slide");                        // {sliding_thresh, half_map_size}: mapSliding after every UpdateVoxelMap (LIVMapper.cpp:430-433)
      if (slide.size() >= 2) { vm.config_setting_.map_sliding_en = true; vm.config_setting_.sliding_thresh = slide[0]; vm.config_setting_.half_map_size = (int)slide[1]; }
      vm.BuildVoxelMap(points_from(rd<double>(dir, "seq_bld_pw"), rd<double>(dir, "seq_bld_var")));
      StatesGroup post = state_from(rd<double>(dir, "seq_state0"));
      std::vector<double> traj;
      size_t off = 0;
      for (size_t f = 0; f < seq_counts.size(); f++) {
        StatesGroup prop = post;                                        // stand-in for the IMU propagation: posterior (+) commanded motion, inflated covariance
        const double *mo = &motion[f * 12];
        for (int r = 0; r < 3; r++) {
          for (int c = 0; c < 3; c++) prop.rot_end[r * 3 + c] = post.rot_end[r * 3] * mo[c] + post.rot_end[r * 3 + 1] * mo[3 + c] + post.rot_end[r * 3 + 2] * mo[6 + c];
          prop.pos_end[r] = post.pos_end[r] + mo[9 + r];
        }
        for (int k = 0; k < LIVO2_DIM_STATE; k++) prop.cov[k * LIVO2_DIM_STATE + k] += qd[k];
        const int n = seq_counts[f];
        vm.feats_down_body_.resize(n);
        std::memcpy(vm.feats_down_body_.data(), &scans[off * 3], (size_t)n * 12); off += n;
        vm.state_ = prop;
        vm.StateEstimation(prop);
        post = vm.state_;
        if (vm.device_map_) {                                           // LIVMapper.cpp:413-424 in one device call
          vm.UpdateVoxelMapFromPosterior();
          if (vm.config_setting_.map_sliding_en) { const int nrm = vm.mapSliding(); std::printf("seq frame %zu: mapSliding removed %d root voxels\n", f, nrm); }
          auto so = state_to(post);
          traj.insert(traj.end(), so.begin(), so.begin() + 12);
          std::printf("seq frame %zu: effct_feat_num_=%d, device map update %.0f us\n", f, vm.effct_feat_num_, vm.last_map_kernel_us_);
          continue;
        }
        // LIVMapper.cpp:413-423: world points (float cloud) and their covariance from the posterior
        M3D RE;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) RE[r * 3 + c] = post.rot_end[r * 3] * vm.extR_[c] + post.rot_end[r * 3 + 1] * vm.extR_[3 + c] + post.rot_end[r * 3 + 2] * vm.extR_[6 + c];
        for (int i = 0; i < n; i++) {
          pointWithVar &pv = vm.pv_list_[i];
          const PointXYZ &p = vm.feats_down_body_[i];
          double pi[3];
          for (int r = 0; r < 3; r++) pi[r] = vm.extR_[r * 3] * p.x + vm.extR_[r * 3 + 1] * p.y + vm.extR_[r * 3 + 2] * p.z + vm.extT_[r];
          for (int r = 0; r < 3; r++) pv.point_w[r] = (double)(float)(post.rot_end[r * 3] * pi[0] + post.rot_end[r * 3 + 1] * pi[1] + post.rot_end[r * 3 + 2] * pi[2] + post.pos_end[r]);
          const M3D &X = vm.cross_mat_list_[i], &Cb = vm.body_cov_list_[i];
          double A[9], B[9];
          for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { A[r * 3 + c] = RE[r * 3] * Cb[c] + RE[r * 3 + 1] * Cb[3 + c] + RE[r * 3 + 2] * Cb[6 + c];
                                                                     B[r * 3 + c] = X[r * 3] * post.cov[0 * 19 + c] + X[r * 3 + 1] * post.cov[1 * 19 + c] + X[r * 3 + 2] * post.cov[2 * 19 + c]; }
          for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
            pv.var[r * 3 + c] = (A[r * 3] * RE[c * 3] + A[r * 3 + 1] * RE[c * 3 + 1] + A[r * 3 + 2] * RE[c * 3 + 2]) + (B[r * 3] * X[c * 3] + B[r * 3 + 1] * X[c * 3 + 1] + B[r * 3 + 2] * X[c * 3 + 2]) +
                                post.cov[(3 + r) * 19 + 3 + c];
        }
        vm.UpdateVoxelMap(vm.pv_list_);
        if (vm.config_setting_.map_sliding_en) { const int nrm = vm.mapSliding(); std::printf("seq frame %zu: mapSliding removed %d root voxels\n", f, nrm); }
        auto so = state_to(post);
        traj.insert(traj.end(), so.begin(), so.begin() + 12);            // rot9 pos3
        std::printf("seq frame %zu: effct_feat_num_=%d, %d plane fits in %d batches\n", f, vm.effct_feat_num_, vm.last_fit_count_, vm.last_fit_rounds_);
      }
      wr(dir, "seq_out_traj", traj.data(), traj.size());
      if (vm.device_map_) dump_device_map(dev, dir, "seq_out_"); else dump_map(vm, dir, "seq_out_");
    }
    // ---- visual -----------------------------------------------------------------------------------------------------------
    auto img = rd<uint8_t>(dir, "img");
    if (!img.empty()) {
      auto cam = rd<double>(dir, "vis_cfg");   // fx fy cx cy w h img_point_cov L max_it exposure Rcl9 Pcl3 extR9 extT3
      auto pos = rd<double>(dir, "vis_pos"); auto warp = rd<float>(dir, "vis_warp"); auto sl = rd<int32_t>(dir, "vis_search"); auto ie = rd<double>(dir, "vis_invexpo");
      VIOManager vio(dev);
      vio.fx = cam[0]; vio.fy = cam[1]; vio.cx = cam[2]; vio.cy = cam[3]; vio.width = (int)cam[4]; vio.height = (int)cam[5];
      vio.img_point_cov = cam[6]; vio.patch_pyrimid_level = (int)cam[7]; vio.max_iterations = (int)cam[8]; vio.exposure_estimate_en = cam[9] != 0;
      M3D Rcl, extR; V3D Pcl, extT;
      for (int k = 0; k < 9; k++) { Rcl[k] = cam[10 + k]; extR[k] = cam[22 + k]; }
      for (int k = 0; k < 3; k++) { Pcl[k] = cam[19 + k]; extT[k] = cam[31 + k]; }
      vio.setImuToLidarExtrinsic(extT, extR); vio.setLidarToCameraExtrinsic(Rcl, Pcl);
      const int M = (int)sl.size(), L = vio.patch_pyrimid_level;
      std::vector<VisualPoint> pts(M);
      SubSparseMap sm;
      for (int i = 0; i < M; i++) {
        for (int k = 0; k < 3; k++) pts[i].pos_[k] = pos[(size_t)i * 3 + k];
        sm.voxel_points.push_back(&pts[i]);
        sm.warp_patch.emplace_back(warp.begin() + (size_t)i * L * 64, warp.begin() + (size_t)(i + 1) * L * 64);
        sm.search_levels.push_back(sl[i]); sm.inv_expo_list.push_back(ie[i]);
      }
      vio.visual_submap = &sm; vio.total_points = M;
      StatesGroup st = state_from(rd<double>(dir, "vis_state_in")), prop = state_from(rd<double>(dir, "vis_state_prop"));
      vio.state = &st; vio.state_propagat = &prop;
      GrayImage g{img.data(), vio.width, vio.height, vio.width};
      vio.kernel_times_en = true;
      vio.computeJacobianAndUpdateEKF(g);                              // <- the reference call site (src/vio.cpp:1810)
      auto so = state_to(st);
      { double tms[2] = {vio.compute_jacobian_time, vio.update_ekf_time}; wr(dir, "vis_out_times", tms, 2); }
      wr(dir, "vis_out_state", so.data(), so.size());
      wr(dir, "vis_out_errors", sm.errors.data(), sm.errors.size());
      wr(dir, "vis_out_G", vio.G.data(), vio.G.size());
      std::printf("visual: M=%d\n", M);
      // ---- retrieveFromVisualSparseMap's per-point tail on the device, then the visual update on the retrieved frame ----
      auto rcfg = rd<double>(dir, "retr_cfg");     // R_cur9 t_cur3 inv_expo_cur normal_en ncc_en ncc_thre outlier_threshold L
      if (!rcfg.empty()) {
        auto rimg = rd<uint8_t>(dir, "retr_img"), rrefs = rd<uint8_t>(dir, "retr_ref_imgs");
        auto rpos = rd<double>(dir, "retr_pos"), rnrm = rd<double>(dir, "retr_normal"), rpx = rd<double>(dir, "retr_ref_px"), rf = rd<double>(dir, "retr_ref_f");
        auto rR = rd<double>(dir, "retr_ref_R"), rt = rd<double>(dir, "retr_ref_t"), rie = rd<double>(dir, "retr_ref_inv_expo");
        auto ridx = rd<int32_t>(dir, "retr_ref_img_idx"), rlvl = rd<int32_t>(dir, "retr_ref_level");
        const int n = (int)ridx.size();
        const size_t bytes = (size_t)vio.width * vio.height;
        std::vector<VisualPoint> rp(n); std::vector<Feature> ft(n); std::vector<VIOManager::Candidate> cands(n);
        for (int i = 0; i < n; i++) {
          for (int k = 0; k < 3; k++) { rp[i].pos_[k] = rpos[(size_t)i * 3 + k]; rp[i].normal_[k] = rnrm[(size_t)i * 3 + k]; ft[i].f_[k] = rf[(size_t)i * 3 + k]; ft[i].t_f_w[k] = rt[(size_t)i * 3 + k]; }
          for (int k = 0; k < 9; k++) ft[i].R_f_w[k] = rR[(size_t)i * 9 + k];
          ft[i].px_ = {rpx[(size_t)i * 2], rpx[(size_t)i * 2 + 1]}; ft[i].img_ = rrefs.data() + bytes * ridx[i]; ft[i].level_ = rlvl[i]; ft[i].inv_expo_time_ = rie[i];
          cands[i] = {&rp[i], &ft[i]};
        }
        for (int k = 0; k < 9; k++) vio.R_f_w_new[k] = rcfg[k];
        for (int k = 0; k < 3; k++) vio.t_f_w_new[k] = rcfg[9 + k];
        StatesGroup st2 = state_from(rd<double>(dir, "retr_state_in")), prop2 = state_from(rd<double>(dir, "retr_state_prop"));
        st2.inv_expo_time = rcfg[12];
        vio.patch_pyrimid_level = (int)rcfg[17];
        vio.state = &st2; vio.state_propagat = &prop2;
        vio.normal_en = rcfg[13] != 0; vio.ncc_en = rcfg[14] != 0; vio.ncc_thre = rcfg[15]; vio.outlier_threshold = rcfg[16];
        SubSparseMap sm2; vio.visual_submap = &sm2;
        GrayImage g2{rimg.data(), vio.width, vio.height, vio.width};
        // selection half first, when the dump carries a visual map: feat_map filled the way insertPointIntoVoxelMap files points (vio.cpp:227-244)
        auto sel_pos = rd<double>(dir, "sel_pos");
        if (!sel_pos.empty()) {
          auto sel_key = rd<int64_t>(dir, "sel_keys"); auto sel_act = rd<uint8_t>(dir, "sel_active"); auto sel_pg = rd<double>(dir, "sel_pg");
          auto sg = rd<double>(dir, "sel_cfg");          // R_cur9 t_cur3 border grid_n_height
          const size_t nv = sel_act.size();
          std::vector<VisualPoint> vp(nv); std::vector<Feature> dummy(1);
          for (size_t i = 0; i < nv; i++) {
            for (int k = 0; k < 3; k++) vp[i].pos_[k] = sel_pos[i * 3 + k];
            if (sel_act[i]) vp[i].obs_.push_back(&dummy[0]);
            const VOXEL_LOCATION key(sel_key[i * 3], sel_key[i * 3 + 1], sel_key[i * 3 + 2]);
            auto it = vio.feat_map.find(key);
            if (it == vio.feat_map.end()) it = vio.feat_map.emplace(key, new VOXEL_POINTS).first;
            it->second->voxel_points.push_back(&vp[i]); it->second->count++;
          }
          M3D Rk = vio.R_f_w_new; V3D tk = vio.t_f_w_new;
          for (int k = 0; k < 9; k++) vio.R_f_w_new[k] = sg[k];
          for (int k = 0; k < 3; k++) vio.t_f_w_new[k] = sg[9 + k];
          vio.border = (int)sg[12]; vio.grid_n_height = (int)sg[13]; vio.grid_size = 5; vio.grid_n_width = 0;
          std::vector<pointWithVar> pg(sel_pg.size() / 3);
          for (size_t i = 0; i < pg.size(); i++) for (int k = 0; k < 3; k++) pg[i].point_w[k] = sel_pg[i * 3 + k];
          auto kept_pts = vio.selectFromVisualSparseMap(pg);
          std::vector<int32_t> kept_idx;
          for (VisualPoint *p : kept_pts) kept_idx.push_back((int32_t)(p - vp.data()));
          wr(dir, "sel_out_kept", kept_idx.data(), kept_idx.size()); wr(dir, "sel_out_map_dist", vio.map_dist.data(), vio.map_dist.size());
          std::printf("select: %zu points kept of %zu grid cells\n", kept_idx.size(), vio.map_dist.size());
          for (auto &kv : vio.feat_map) delete kv.second;
          vio.feat_map.clear(); vio.feat_map_dirty_ = true;
          vio.R_f_w_new = Rk; vio.t_f_w_new = tk;
        }
        vio.warpAndGateCandidates(g2, cands);
        std::vector<int32_t> kept;
        for (auto *p : sm2.voxel_points) kept.push_back((int32_t)(p - rp.data()));
        wr(dir, "retr_out_kept", kept.data(), kept.size()); wr(dir, "retr_out_errors", sm2.errors.data(), sm2.errors.size());
        std::vector<int32_t> sl2(sm2.search_levels.begin(), sm2.search_levels.end());
        wr(dir, "retr_out_search", sl2.data(), sl2.size());
        vio.computeJacobianAndUpdateEKF(g2);
        auto so2 = state_to(st2);
        wr(dir, "retr_out_state", so2.data(), so2.size());
        std::printf("retrieve: %d of %d candidates kept\n", vio.total_points, n);
      }
      // ---- the whole retrieveFromVisualSparseMap through the shim (selection -> reference-patch choice -> tail), then the visual update ----
      auto ccfg = rd<double>(dir, "chain_cfg");    // R_cur9 t_cur3 inv_expo_cur normal_en ncc_en ncc_thre outlier_threshold L border grid_n_height
      if (!ccfg.empty()) {
        auto cimg = rd<uint8_t>(dir, "chain_img"), crefs = rd<uint8_t>(dir, "chain_ref_imgs");
        auto cpg = rd<double>(dir, "chain_pg"), cpos = rd<double>(dir, "chain_pos"), cnrm = rd<double>(dir, "chain_normal");
        auto ckey = rd<int64_t>(dir, "chain_keys"); auto cninit = rd<uint8_t>(dir, "chain_ninit"); auto crefp = rd<int32_t>(dir, "chain_ref_patch");
        auto ooff = rd<int32_t>(dir, "chain_obs_offset"), oid = rd<int32_t>(dir, "chain_obs_id"), oimg = rd<int32_t>(dir, "chain_obs_img_idx"), olvl = rd<int32_t>(dir, "chain_obs_level");
        auto opx = rd<double>(dir, "chain_obs_px"), of = rd<double>(dir, "chain_obs_f"), oR = rd<double>(dir, "chain_obs_R"), ot = rd<double>(dir, "chain_obs_t"), oie = rd<double>(dir, "chain_obs_inv_expo");
        auto opatch = rd<float>(dir, "chain_obs_patch");
        const size_t nv = cninit.size(), no = oid.size(), bytes = (size_t)vio.width * vio.height;
        std::vector<VisualPoint> vp(nv); std::vector<Feature> ft(no);
        for (size_t k = 0; k < no; k++) {
          ft[k].id_ = oid[k]; ft[k].patch_ = opatch.data() + 64 * k; ft[k].img_ = crefs.data() + bytes * oimg[k]; ft[k].px_ = {opx[k * 2], opx[k * 2 + 1]};
          ft[k].level_ = olvl[k]; ft[k].inv_expo_time_ = oie[k];
          for (int j = 0; j < 3; j++) { ft[k].f_[j] = of[k * 3 + j]; ft[k].t_f_w[j] = ot[k * 3 + j]; }
          for (int j = 0; j < 9; j++) ft[k].R_f_w[j] = oR[k * 9 + j];
        }
        for (size_t i = 0; i < nv; i++) {
          for (int k = 0; k < 3; k++) { vp[i].pos_[k] = cpos[i * 3 + k]; vp[i].normal_[k] = cnrm[i * 3 + k]; }
          vp[i].is_normal_initialized_ = cninit[i] != 0;
          for (int k = ooff[i]; k < ooff[i + 1]; k++) vp[i].obs_.push_back(&ft[k]);
          if (crefp[i] >= 0) { vp[i].ref_patch = &ft[crefp[i]]; vp[i].has_ref_patch_ = true; }
          const VOXEL_LOCATION key(ckey[i * 3], ckey[i * 3 + 1], ckey[i * 3 + 2]);
          auto it = vio.feat_map.find(key);
          if (it == vio.feat_map.end()) it = vio.feat_map.emplace(key, new VOXEL_POINTS).first;
          it->second->voxel_points.push_back(&vp[i]); it->second->count++;
        }
        vio.feat_map_dirty_ = true;
        for (int k = 0; k < 9; k++) vio.R_f_w_new[k] = ccfg[k];
        for (int k = 0; k < 3; k++) vio.t_f_w_new[k] = ccfg[9 + k];
        StatesGroup st3 = state_from(rd<double>(dir, "chain_state_in")), prop3 = state_from(rd<double>(dir, "chain_state_prop"));
        st3.inv_expo_time = ccfg[12];
        vio.normal_en = ccfg[13] != 0; vio.ncc_en = ccfg[14] != 0; vio.ncc_thre = ccfg[15]; vio.outlier_threshold = ccfg[16]; vio.patch_pyrimid_level = (int)ccfg[17];
        vio.border = (int)ccfg[18]; vio.grid_n_height = (int)ccfg[19]; vio.grid_size = 5; vio.grid_n_width = 0;
        vio.state = &st3; vio.state_propagat = &prop3;
        SubSparseMap sm3; vio.visual_submap = &sm3;
        GrayImage g3{cimg.data(), vio.width, vio.height, vio.width};
        std::vector<pointWithVar> pg(cpg.size() / 3);
        for (size_t i = 0; i < pg.size(); i++) for (int k = 0; k < 3; k++) pg[i].point_w[k] = cpg[i * 3 + k];
        vio.retrieveFromVisualSparseMap(g3, pg);                        // <- the reference call site (src/vio.cpp:1808)
        std::vector<int32_t> kept, kept_obs, refp(nv, -1), sl3(sm3.search_levels.begin(), sm3.search_levels.end());
        for (size_t k = 0; k < sm3.voxel_points.size(); k++) { kept.push_back((int32_t)(sm3.voxel_points[k] - vp.data())); kept_obs.push_back((int32_t)(sm3.voxel_points[k]->ref_patch ? sm3.voxel_points[k]->ref_patch - ft.data() : -1)); }
        for (size_t i = 0; i < nv; i++) if (vp[i].has_ref_patch_) refp[i] = (int32_t)(vp[i].ref_patch - ft.data());
        wr(dir, "chain_out_kept", kept.data(), kept.size()); wr(dir, "chain_out_kept_ref", kept_obs.data(), kept_obs.size());
        wr(dir, "chain_out_errors", sm3.errors.data(), sm3.errors.size()); wr(dir, "chain_out_search", sl3.data(), sl3.size());
        wr(dir, "chain_out_inv_expo", sm3.inv_expo_list.data(), sm3.inv_expo_list.size());
        wr(dir, "chain_out_ref_patch", refp.data(), refp.size()); wr(dir, "chain_out_map_dist", vio.map_dist.data(), vio.map_dist.size());
        vio.computeJacobianAndUpdateEKF(g3);
        auto so3 = state_to(st3);
        wr(dir, "chain_out_state", so3.data(), so3.size());
        std::printf("retrieveFromVisualSparseMap: %d points in the sub-map (%zu visual points, %zu observations)\n", vio.total_points, nv, no);
        for (auto &kv : vio.feat_map) delete kv.second;
        vio.feat_map.clear();
      }
    }
  } catch (const std::exception &e) { std::fprintf(stderr, "shim_demo: %s\n", e.what()); return 1; }
  return 0;
}
