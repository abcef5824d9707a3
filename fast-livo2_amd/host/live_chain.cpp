// One chained live LIO + VIO frame through the C++ host shim, timed: what LIVMapper::handleLIO (src/LIVMapper.cpp:336-482) and handleVIO (281-334) do per frame,
//   StateEstimation(state_propagat)            livo2 device tree resident (VoxelMapManager::device_map_)
//   UpdateVoxelMapFromPosterior()              LIVMapper.cpp:413-424 + UpdateVoxelMap, on the device
//   retrieveFromVisualSparseMap(img, pg)       vio.cpp:352-782, feat_map mirrored on the device
//   computeJacobianAndUpdateEKF(img)           vio.cpp:784-802
// called back to back on the reference's own containers (feats_down_body_, state_, visual_submap, *state), F frames after a warm-up frame.
// Inputs: the dump directory of scenarios/live_inputs.py (seq_* = map + scans + motion, chain_* = visual map + image).  The LiDAR world and the visual world
// of this synthetic input are two scenes (the visual retrieval uses its own scan points), so the poses of the two halves are not chained — the CALLS are.
// Usage: live_chain <dir> [lean]   -> prints ms per frame and per stage; writes live_out.bin = [F][4] stage milliseconds
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

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
static StatesGroup state_from(const std::vector<double> &v) { StatesGroup s; livo2_state a; std::memcpy(&a, v.data(), sizeof(a)); s.from_abi(a); return s; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: live_chain <dir>\n"); return 2; }
  const std::string dir = argv[1];
  try {
    Device dev(0);
    // ---- LiDAR side: device-resident VoxelMap from a first sweep, then the frames' scans
    auto seq_counts = rd<int32_t>(dir, "seq_counts");
    auto mc = rd<double>(dir, "seq_map_cfg"); auto lc = rd<double>(dir, "seq_lidar_cfg");
    auto scans = rd<float>(dir, "seq_scans"); auto motion = rd<double>(dir, "seq_motion"); auto qd = rd<double>(dir, "seq_q");
    if (seq_counts.empty() || mc.empty() || lc.empty()) { std::fprintf(stderr, "live_chain: seq_* inputs missing\n"); return 2; }
    VoxelMapManager vm(dev);
    vm.device_map_ = true;
    vm.host_point_lists_ = !(argc > 2 && std::string(argv[2]) == "lean");       // "lean": no pv_list_ / ptpl_list_ on the host (their consumers run on the device)
    vm.config_setting_.max_voxel_size_ = mc[0]; vm.config_setting_.max_layer_ = (int)mc[1]; vm.config_setting_.max_points_num_ = (int)mc[2]; vm.config_setting_.planner_threshold_ = mc[3];
    vm.config_setting_.layer_init_num_.assign(5, 5);
    for (int k = 0; k < 5; k++) vm.config_setting_.layer_init_num_[k] = (int)mc[4 + k];
    vm.config_setting_.max_iterations_ = (int)lc[0]; vm.config_setting_.sigma_num_ = lc[2]; vm.config_setting_.dept_err_ = lc[3]; vm.config_setting_.beam_err_ = lc[4];
    for (int k = 0; k < 9; k++) vm.extR_[k] = lc[6 + k];
    for (int k = 0; k < 3; k++) vm.extT_[k] = lc[15 + k];
    {
      auto pw = rd<double>(dir, "seq_bld_pw"), var = rd<double>(dir, "seq_bld_var");
      std::vector<pointWithVar> pts(pw.size() / 3);
      for (size_t i = 0; i < pts.size(); i++) { for (int k = 0; k < 3; k++) pts[i].point_w[k] = pw[i * 3 + k]; for (int k = 0; k < 9; k++) pts[i].var[k] = var[i * 9 + k]; }
      vm.device_map_max_roots_ = (int)std::max<size_t>(20000, pts.size() / 4);
      vm.BuildVoxelMap(pts);
    }
    // ---- visual side: feat_map of VisualPoints with their observations, the current image
    auto cam = rd<double>(dir, "vis_cfg");   // fx fy cx cy w h img_point_cov L max_it exposure Rcl9 Pcl3 extR9 extT3
    auto ccfg = rd<double>(dir, "chain_cfg");    // R_cur9 t_cur3 inv_expo_cur normal_en ncc_en ncc_thre outlier_threshold L border grid_n_height
    if (cam.empty() || ccfg.empty()) { std::fprintf(stderr, "live_chain: vis_cfg / chain_* inputs missing\n"); return 2; }
    VIOManager vio(dev);
    vio.fx = cam[0]; vio.fy = cam[1]; vio.cx = cam[2]; vio.cy = cam[3]; vio.width = (int)cam[4]; vio.height = (int)cam[5];
    vio.img_point_cov = cam[6]; vio.patch_pyrimid_level = (int)cam[7]; vio.max_iterations = (int)cam[8]; vio.exposure_estimate_en = cam[9] != 0;
    M3D Rcl, extR; V3D Pcl, extT;
    for (int k = 0; k < 9; k++) { Rcl[k] = cam[10 + k]; extR[k] = cam[22 + k]; }
    for (int k = 0; k < 3; k++) { Pcl[k] = cam[19 + k]; extT[k] = cam[31 + k]; }
    vio.setImuToLidarExtrinsic(extT, extR); vio.setLidarToCameraExtrinsic(Rcl, Pcl);
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
    vio.normal_en = ccfg[13] != 0; vio.ncc_en = ccfg[14] != 0; vio.ncc_thre = ccfg[15]; vio.outlier_threshold = ccfg[16]; vio.patch_pyrimid_level = (int)ccfg[17];
    vio.border = (int)ccfg[18]; vio.grid_n_height = (int)ccfg[19]; vio.grid_size = 5; vio.grid_n_width = 0;
    const StatesGroup vst0 = state_from(rd<double>(dir, "chain_state_in")), vprop0 = state_from(rd<double>(dir, "chain_state_prop"));
    GrayImage g3{cimg.data(), vio.width, vio.height, vio.width};
    std::vector<pointWithVar> pg(cpg.size() / 3);
    for (size_t i = 0; i < pg.size(); i++) for (int k = 0; k < 3; k++) pg[i].point_w[k] = cpg[i * 3 + k];

    // ---- the frames
    StatesGroup post = state_from(rd<double>(dir, "seq_state0"));
    const size_t F = seq_counts.size();
    std::vector<double> stage(F * 4, 0.0);
    size_t off = 0;
    double total = 0.0; size_t timed = 0, sub_pts = 0, eff = 0;
    for (size_t f = 0; f < F; f++) {
      StatesGroup prop = post;                                          // stand-in for the IMU propagation: posterior (+) commanded motion, inflated covariance
      const double *mo = &motion[f * 12];
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) prop.rot_end[r * 3 + c] = post.rot_end[r * 3] * mo[c] + post.rot_end[r * 3 + 1] * mo[3 + c] + post.rot_end[r * 3 + 2] * mo[6 + c];
        prop.pos_end[r] = post.pos_end[r] + mo[9 + r];
      }
      for (int k = 0; k < LIVO2_DIM_STATE; k++) prop.cov[k * LIVO2_DIM_STATE + k] += qd[k];
      const int n = seq_counts[f];
      StatesGroup vst = vst0, vprop = vprop0; vst.inv_expo_time = ccfg[12];
      SubSparseMap sm; vio.visual_submap = &sm; vio.state = &vst; vio.state_propagat = &vprop;
      const auto t0 = std::chrono::steady_clock::now();
      vm.feats_down_body_.resize(n);
      std::memcpy(vm.feats_down_body_.data(), &scans[off * 3], (size_t)n * 12); off += n;
      vm.state_ = prop;
      vm.StateEstimation(prop);                                         // LIVMapper.cpp:370
      post = vm.state_;
      const double a = ms_since(t0);
      vm.UpdateVoxelMapFromPosterior();                                 // LIVMapper.cpp:413-424
      const double b = ms_since(t0);
      vio.retrieveFromVisualSparseMap(g3, pg);                          // vio.cpp:1808
      const double c = ms_since(t0);
      vio.computeJacobianAndUpdateEKF(g3);                              // vio.cpp:1810
      const double d = ms_since(t0);
      stage[f * 4] = a; stage[f * 4 + 1] = b - a; stage[f * 4 + 2] = c - b; stage[f * 4 + 3] = d - c;
      if (f >= 1) { total += d; timed++; sub_pts += (size_t)vio.total_points; eff += (size_t)vm.effct_feat_num_; }      // frame 0 = warm-up (allocations, feat_map mirror)
    }
    {
      std::ofstream o(dir + "/live_out.bin", std::ios::binary);
      o.write((const char *)stage.data(), stage.size() * 8);
    }
    double s[4] = {0, 0, 0, 0};
    for (size_t f = 1; f < F; f++) for (int k = 0; k < 4; k++) s[k] += stage[f * 4 + k];
    const double T = timed ? (double)timed : 1.0;
    std::printf("live_chain%s: %zu frames timed, %.3f ms per frame (StateEstimation %.3f, UpdateVoxelMapFromPosterior %.3f, retrieveFromVisualSparseMap %.3f, computeJacobianAndUpdateEKF %.3f); "
                "mean scan %.0f points, effct_feat_num_ %.0f, sub-map %.0f patches\n",
                vm.host_point_lists_ ? "" : " (lean)", timed, total / T, s[0] / T, s[1] / T, s[2] / T, s[3] / T, (double)off / (double)F, (double)eff / T, (double)sub_pts / T);
    for (auto &kv : vio.feat_map) delete kv.second;
  } catch (const std::exception &e) { std::fprintf(stderr, "live_chain: %s\n", e.what()); return 1; }
  return 0;
}
