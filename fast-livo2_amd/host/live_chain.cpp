// Chained live LIO + VIO frames through the C++ host shim — ONE scene, the data flow of the reference (all file:line relative to /root/reference):
//   processImu      state_propagat = _state (+ commanded motion: stand-in for the IMU propagation)         src/LIVMapper.cpp:250-257
//   handleLIO       StateEstimation(state_propagat); _state = state_                                       src/LIVMapper.cpp:370-371
//                   UpdateVoxelMap with pv_list_ at the POSTERIOR; the next frame reads THAT map            src/LIVMapper.cpp:413-426
//   handleVIO       state = &_state (shared), state_propagat = _state                                      src/LIVMapper.cpp:135-136, 256
//                   updateFrameState(*state); retrieveFromVisualSparseMap(img, _pv_list); computeJacobianAndUpdateEKF(img)   src/vio.cpp:1799-1810
//   next frame      propagated from the VIO posterior.
// Inputs: the dump directory of scenarios/live_inputs.py (seq_* = first sweep + scans + motion, chain<k>_* = the visual map and image of frame k: points on the
// surfaces of the same room, seen from frame k's true pose).  The visual map of frame k is installed in feat_map before the frame starts (its maintenance —
// generateVisualMapPoints / updateVisualMapPoints / updateReferencePatch — is out of scope) and mirrored on the device by syncFeatMap: that cost is
// reported on its own, outside the four stages.
// Growing-map mode (the directory holds grow_cfg.bin, scenarios/live_inputs.py make_live(grow=n)): ONE visual map installed before frame 0 and changed after every frame
// by a scripted stand-in for the maintenance; syncFeatMap then applies O(changes) deltas and is timed INSIDE the frame (stage 5 of live_out.bin).
// Usage: live_chain <dir> [lean]   "lean": host_point_lists_ = false (no pv_list_ / ptpl_list_ on the host), `pg` read where the map update left it on the GPU, the map
//        update on the context's second stream (LIVO2_LIVE_SYNC_MAP=1: synchronous) and retrieval + visual update through VIOManager::retrieveAndUpdate (LIVO2_LIVE_SPLIT_VIO=1: two calls).
// Output: live_out.bin [F][5] ms (StateEstimation, UpdateVoxelMapFromPosterior, retrieveFromVisualSparseMap, computeJacobianAndUpdateEKF, syncFeatMap),
//         live_states.bin [F][2] livo2_state (LIO posterior, VIO posterior), live_counts.bin [F][2] int32 (effct_feat_num_, total_points),
//         live_sub_pos.bin: pos_ of visual_submap->voxel_points, frame after frame.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "livo2_host.hpp"

using namespace livo2;

template <typename T> static std::vector<T> rd(const std::string &dir, const std::string &name) {
  std::ifstream f(dir + "/" + name + ".bin", std::ios::binary | std::ios::ate);
  if (!f) return {};
  size_t bytes = (size_t)f.tellg(); f.seekg(0);
  std::vector<T> v(bytes / sizeof(T));
  f.read((char *)v.data(), bytes);
  return v;
}
static StatesGroup state_from(const std::vector<double> &v) { StatesGroup s; livo2_state a; std::memcpy(&a, v.data(), sizeof(a)); s.from_abi(a); return s; }
static double ms_since(std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }

struct VisualMapFrame {                      // the arrays of one chain<k>_* dump; feat_map points into them
  std::vector<uint8_t> img, refs, ninit;
  std::vector<double> pos, nrm, opx, of, oR, ot, oie;
  std::vector<int64_t> key;
  std::vector<int32_t> refp, ooff, oid, oimg, olvl;
  std::vector<float> opatch;
  std::vector<VisualPoint> vp;
  std::vector<Feature> ft;
};

static bool load_visual_map(const std::string &dir, int k, size_t img_bytes, VisualMapFrame &m) {
  const std::string p = "chain" + std::to_string(k) + "_";
  m.img = rd<uint8_t>(dir, p + "img"); m.refs = rd<uint8_t>(dir, p + "ref_imgs"); m.ninit = rd<uint8_t>(dir, p + "ninit");
  m.pos = rd<double>(dir, p + "pos"); m.nrm = rd<double>(dir, p + "normal"); m.key = rd<int64_t>(dir, p + "keys"); m.refp = rd<int32_t>(dir, p + "ref_patch");
  m.ooff = rd<int32_t>(dir, p + "obs_offset"); m.oid = rd<int32_t>(dir, p + "obs_id"); m.oimg = rd<int32_t>(dir, p + "obs_img_idx"); m.olvl = rd<int32_t>(dir, p + "obs_level");
  m.opx = rd<double>(dir, p + "obs_px"); m.of = rd<double>(dir, p + "obs_f"); m.oR = rd<double>(dir, p + "obs_R"); m.ot = rd<double>(dir, p + "obs_t");
  m.oie = rd<double>(dir, p + "obs_inv_expo"); m.opatch = rd<float>(dir, p + "obs_patch");
  if (m.img.size() != img_bytes || m.ninit.empty() || m.ooff.size() != m.ninit.size() + 1) return false;
  const size_t nv = m.ninit.size(), no = m.oid.size();
  m.vp.assign(nv, VisualPoint()); m.ft.assign(no, Feature());
  for (size_t i = 0; i < no; i++) {
    Feature &f = m.ft[i];
    f.id_ = m.oid[i]; f.patch_ = m.opatch.data() + 64 * i; f.img_ = m.refs.data() + img_bytes * m.oimg[i]; f.px_ = {m.opx[i * 2], m.opx[i * 2 + 1]};
    f.level_ = m.olvl[i]; f.inv_expo_time_ = m.oie[i];
    for (int j = 0; j < 3; j++) { f.f_[j] = m.of[i * 3 + j]; f.t_f_w[j] = m.ot[i * 3 + j]; }
    for (int j = 0; j < 9; j++) f.R_f_w[j] = m.oR[i * 9 + j];
  }
  for (size_t i = 0; i < nv; i++) {
    for (int j = 0; j < 3; j++) { m.vp[i].pos_[j] = m.pos[i * 3 + j]; m.vp[i].normal_[j] = m.nrm[i * 3 + j]; }
    m.vp[i].is_normal_initialized_ = m.ninit[i] != 0;
    for (int j = m.ooff[i]; j < m.ooff[i + 1]; j++) m.vp[i].obs_.push_back(&m.ft[j]);
    if (m.refp[i] >= 0) { m.vp[i].ref_patch = &m.ft[m.refp[i]]; m.vp[i].has_ref_patch_ = true; }
  }
  return true;
}

// ---- growing-map mode: ONE visual map, changed after every frame by a scripted stand-in for generateVisualMapPoints / updateVisualMapPoints / updateReferencePatch
// (scenarios/visual_map_growth.py; the maintenance logic is out of scope, its effect on the containers is what the mirror has to follow).  The script is replayed
// on the shim's objects through the three hooks a maintainer adds to those functions — insertPointIntoVoxelMap, markPointDirty, erasePointFromVoxelMap — and the
// next syncFeatMap brings the device mirror up to date in O(changes) (livo2_visual_map_apply).
struct GrowScript {
  std::vector<double> new_pos, new_normal, opx, of, oR, ot, oie;
  std::vector<int32_t> oid, oimg, olvl, t_point, t_pop, t_push, t_ref, t_flip, t_toggle, t_remove;
  std::vector<float> opatch;
  std::vector<uint8_t> ref_img;
  std::vector<Feature> ft;                  // storage of the new Features / points of this script (stable: reserved before use)
  std::vector<VisualPoint> vp;
};
static bool load_grow_script(const std::string &dir, int k, GrowScript &g) {
  const std::string p = "grow" + std::to_string(k) + "_";
  g.new_pos = rd<double>(dir, p + "new_pos"); g.new_normal = rd<double>(dir, p + "new_normal"); g.ref_img = rd<uint8_t>(dir, p + "ref_img");
  g.oid = rd<int32_t>(dir, p + "obs_id"); g.oimg = rd<int32_t>(dir, p + "obs_img_idx"); g.olvl = rd<int32_t>(dir, p + "obs_level"); g.opx = rd<double>(dir, p + "obs_px");
  g.of = rd<double>(dir, p + "obs_f"); g.oR = rd<double>(dir, p + "obs_R"); g.ot = rd<double>(dir, p + "obs_t"); g.oie = rd<double>(dir, p + "obs_inv_expo"); g.opatch = rd<float>(dir, p + "obs_patch");
  g.t_point = rd<int32_t>(dir, p + "t_point"); g.t_pop = rd<int32_t>(dir, p + "t_pop"); g.t_push = rd<int32_t>(dir, p + "t_push"); g.t_ref = rd<int32_t>(dir, p + "t_ref");
  g.t_flip = rd<int32_t>(dir, p + "t_flip"); g.t_toggle = rd<int32_t>(dir, p + "t_toggle"); g.t_remove = rd<int32_t>(dir, p + "t_remove");
  return !g.ref_img.empty() && g.oid.size() * 64 == g.opatch.size() && g.t_point.size() == g.t_ref.size();
}
// replays one script: `points` / `feats` = file index -> object (the scripts count points and observations the way scenarios/visual_map_growth.py does)
static void replay_grow_script(VIOManager &vio, GrowScript &g, std::vector<VisualPoint *> &points, std::vector<Feature *> &feats) {
  const size_t no = g.oid.size(), nn = g.new_pos.size() / 3;
  g.ft.assign(no, Feature()); g.vp.assign(nn, VisualPoint());
  for (size_t i = 0; i < no; i++) {
    Feature &f = g.ft[i];
    f.id_ = g.oid[i]; f.patch_ = g.opatch.data() + 64 * i; f.img_ = g.ref_img.data(); f.px_ = {g.opx[i * 2], g.opx[i * 2 + 1]}; f.level_ = g.olvl[i]; f.inv_expo_time_ = g.oie[i];
    for (int j = 0; j < 3; j++) { f.f_[j] = g.of[i * 3 + j]; f.t_f_w[j] = g.ot[i * 3 + j]; }
    for (int j = 0; j < 9; j++) f.R_f_w[j] = g.oR[i * 9 + j];
    feats.push_back(&f);
  }
  const size_t m0 = feats.size() - no;
  for (size_t k = 0; k < nn; k++) {                                                  // generateVisualMapPoints: a new point with its one Feature (global index m0 + k)
    VisualPoint &v = g.vp[k];
    for (int j = 0; j < 3; j++) { v.pos_[j] = g.new_pos[k * 3 + j]; v.normal_[j] = g.new_normal[k * 3 + j]; }
    v.is_normal_initialized_ = true; v.obs_.push_back(feats[m0 + k]);
    vio.insertPointIntoVoxelMap(&v);
    points.push_back(&v);
  }
  for (size_t q = 0; q < g.t_point.size(); q++) {                                    // updateVisualMapPoints / updateReferencePatch on resident points
    VisualPoint *pt = points[g.t_point[q]];
    if (g.t_remove[q]) { vio.erasePointFromVoxelMap(pt); pt->obs_.clear(); pt->ref_patch = nullptr; pt->has_ref_patch_ = false; continue; }
    if (g.t_pop[q]) {                                                                // deleteFeatureRef (visual_point.cpp:40-55)
      Feature *victim = pt->obs_.back(); pt->obs_.pop_back();
      if (pt->ref_patch == victim) { pt->ref_patch = nullptr; pt->has_ref_patch_ = false; }
    }
    if (g.t_push[q] >= 0) pt->obs_.insert(pt->obs_.begin(), feats[g.t_push[q]]);    // addFrameRef: push_front
    if (g.t_ref[q] == -1) { pt->ref_patch = nullptr; pt->has_ref_patch_ = false; }
    else if (g.t_ref[q] >= 0) { pt->ref_patch = feats[g.t_ref[q]]; pt->has_ref_patch_ = true; }
    if (g.t_flip[q]) for (int j = 0; j < 3; j++) pt->normal_[j] = -pt->normal_[j];
    if (g.t_toggle[q]) pt->is_normal_initialized_ = !pt->is_normal_initialized_;
    vio.markPointDirty(pt);
  }
}

static void install_visual_map(VIOManager &vio, VisualMapFrame &m) {
  for (auto &kv : vio.feat_map) delete kv.second;
  vio.feat_map.clear();
  for (size_t i = 0; i < m.vp.size(); i++) {
    const VOXEL_LOCATION key(m.key[i * 3], m.key[i * 3 + 1], m.key[i * 3 + 2]);
    auto it = vio.feat_map.find(key);
    if (it == vio.feat_map.end()) it = vio.feat_map.emplace(key, new VOXEL_POINTS).first;
    it->second->voxel_points.push_back(&m.vp[i]); it->second->count++;
  }
  vio.feat_map_dirty_ = true;
}

int main(int argc, char **argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: live_chain <dir> [lean]\n"); return 2; }
  const std::string dir = argv[1];
  const bool lean = argc > 2 && std::string(argv[2]) == "lean";
  try {
    Device dev(0);
    // ---- LiDAR side: device-resident VoxelMap from a first sweep, then the frames' scans
    auto seq_counts = rd<int32_t>(dir, "seq_counts");
    auto mc = rd<double>(dir, "seq_map_cfg"); auto lc = rd<double>(dir, "seq_lidar_cfg");
    auto scans = rd<float>(dir, "seq_scans"); auto motion = rd<double>(dir, "seq_motion"); auto qd = rd<double>(dir, "seq_q");
    if (seq_counts.empty() || mc.empty() || lc.empty()) { std::fprintf(stderr, "live_chain: seq_* inputs missing\n"); return 2; }
    VoxelMapManager vm(dev);
    vm.device_map_ = true;
    vm.host_point_lists_ = !lean;
    vm.async_map_update_ = std::getenv("LIVO2_LIVE_SYNC_MAP") == nullptr;       // lean: the map update runs beside handleVIO on the context's second stream (round 6)
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
    // ---- visual side
    auto cam = rd<double>(dir, "vis_cfg");       // fx fy cx cy w h img_point_cov L max_it exposure Rcl9 Pcl3 extR9 extT3
    auto ccfg = rd<double>(dir, "chain_cfg");    // n_frames normal_en ncc_en ncc_thre outlier_threshold L border grid_n_height
    if (cam.empty() || ccfg.size() < 8) { std::fprintf(stderr, "live_chain: vis_cfg / chain_cfg inputs missing\n"); return 2; }
    VIOManager vio(dev);
    vio.fx = cam[0]; vio.fy = cam[1]; vio.cx = cam[2]; vio.cy = cam[3]; vio.width = (int)cam[4]; vio.height = (int)cam[5];
    vio.img_point_cov = cam[6]; vio.max_iterations = (int)cam[8]; vio.exposure_estimate_en = cam[9] != 0;
    M3D Rcl, extR; V3D Pcl, extT;
    for (int k = 0; k < 9; k++) { Rcl[k] = cam[10 + k]; extR[k] = cam[22 + k]; }
    for (int k = 0; k < 3; k++) { Pcl[k] = cam[19 + k]; extT[k] = cam[31 + k]; }
    vio.setImuToLidarExtrinsic(extT, extR); vio.setLidarToCameraExtrinsic(Rcl, Pcl);
    vio.normal_en = ccfg[1] != 0; vio.ncc_en = ccfg[2] != 0; vio.ncc_thre = ccfg[3]; vio.outlier_threshold = ccfg[4]; vio.patch_pyrimid_level = (int)ccfg[5];
    vio.border = (int)ccfg[6]; vio.grid_n_height = (int)ccfg[7]; vio.grid_size = 5; vio.grid_n_width = 0;
    vio.pg_from_map_update_ = lean;
    const bool pair_call = std::getenv("LIVO2_LIVE_SPLIT_VIO") == nullptr;   // lean: VIOManager::retrieveAndUpdate (round 6); the env restores the two separate calls
    const size_t img_bytes = (size_t)vio.width * vio.height;

    // ---- the frames
    StatesGroup state = state_from(rd<double>(dir, "seq_state0"));        // LIVMapper::_state
    StatesGroup state_propagat;                                            // LIVMapper::state_propagat
    vio.state = &state; vio.state_propagat = &state_propagat;              // LIVMapper.cpp:135-136
    const size_t F = seq_counts.size();
    if ((size_t)ccfg[0] != F) { std::fprintf(stderr, "live_chain: %zu scans but %d visual maps\n", F, (int)ccfg[0]); return 2; }
    std::vector<double> stage(F * 5, 0.0), states(F * 2 * sizeof(livo2_state) / 8, 0.0), sub_pos;
    std::vector<int32_t> counts(F * 2, 0);
    size_t off = 0;
    double total = 0.0; size_t timed = 0, sub_pts = 0, eff = 0;
    VisualMapFrame vmap;
    const size_t warm = F >= 4 ? 2 : (F >= 2 ? 1 : 0);
    // growing-map mode (grow_cfg present): chain0_* is installed once, grow<k>_* scripts change it after frame k, grow<k>_img is the image of frame k >= 1
    const auto grow_cfg = rd<int32_t>(dir, "grow_cfg");
    const bool grow = !grow_cfg.empty();
    std::vector<GrowScript> scripts(grow ? (size_t)grow_cfg[0] : 0);
    std::vector<VisualPoint *> g_points; std::vector<Feature *> g_feats;
    std::vector<uint8_t> cur_img;
    for (size_t f = 0; f < F; f++) {
      if (!grow || f == 0) {
        if (!load_visual_map(dir, (int)f, img_bytes, vmap)) { std::fprintf(stderr, "live_chain: chain%zu_* inputs missing\n", f); return 2; }
        install_visual_map(vio, vmap);
        if (grow) { for (auto &v : vmap.vp) g_points.push_back(&v); for (auto &x : vmap.ft) g_feats.push_back(&x); }
      }
      if (grow && f > 0) { cur_img = rd<uint8_t>(dir, "grow" + std::to_string(f) + "_img"); if (cur_img.size() != img_bytes) { std::fprintf(stderr, "live_chain: grow%zu_img missing\n", f); return 2; } }
      GrayImage img{(grow && f > 0) ? cur_img.data() : vmap.img.data(), vio.width, vio.height, vio.width};
      const auto tm = std::chrono::steady_clock::now();
      vio.syncFeatMap(img);
      dev.check(livo2_ctx_synchronize(dev.ctx()));                                   // (the mirror's uploads are asynchronous: their tail must not be billed to StateEstimation)
      stage[f * 5 + 4] = ms_since(tm);
      // Installing a whole new visual map is 30-150 ms of HOST work per frame in this program (the reference changes its map incrementally): long enough for the GPU to
      // drop into an idle power state, and the first device operation afterwards then waits 10-20 ms for it to come back (seen as StateEstimation 15-28 ms on some
      // frames of the C4-sized chain, with every kernel of the update taking its usual 20 us in the rocprofv3 trace).  One tiny device round trip absorbs that here.
      { int32_t counts[8]; for (int k = 0; k < 3; k++) dev.check(livo2_map_tree_stats(dev.ctx(), counts)); }
      // processImu: the propagated state (stand-in: posterior (+) commanded motion, inflated covariance)
      const double *mo = &motion[f * 12];
      StatesGroup prop = state;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) prop.rot_end[r * 3 + c] = state.rot_end[r * 3] * mo[c] + state.rot_end[r * 3 + 1] * mo[3 + c] + state.rot_end[r * 3 + 2] * mo[6 + c];
        prop.pos_end[r] = state.pos_end[r] + mo[9 + r];
      }
      for (int k = 0; k < LIVO2_DIM_STATE; k++) prop.cov[k * LIVO2_DIM_STATE + k] += qd[k];
      state = prop; state_propagat = state; vm.state_ = state;                       // LIVMapper.cpp:256-257
      const int n = seq_counts[f];
      SubSparseMap sm; vio.visual_submap = &sm;
      const auto t0 = std::chrono::steady_clock::now();
      vm.feats_down_body_.resize(n);
      std::memcpy(vm.feats_down_body_.data(), &scans[off * 3], (size_t)n * 12); off += n;
      vm.StateEstimation(state_propagat);                                            // LIVMapper.cpp:370
      state = vm.state_;                                                             // LIVMapper.cpp:371
      const double a = ms_since(t0);
      vm.UpdateVoxelMapFromPosterior();                                              // LIVMapper.cpp:413-424 (pv_list_ at the posterior + UpdateVoxelMap)
      const double b = ms_since(t0);
      livo2_state s_lio; state.to_abi(s_lio);
      state_propagat = state;                                                        // processImu before the VIO step (LIVMapper.cpp:256)
      vio.updateFrameState(state);                                                   // vio.cpp:1799-1800
      double c;
      if (pair_call) {                                                               // vio.cpp:1808 + 1810 as one member: the update runs on the GPU while the host builds visual_submap's lists
        vio.retrieveAndUpdate(img, vm.pv_list_);
        c = vio.total_points > 0 ? std::chrono::duration<double, std::milli>(vio.update_enqueued_at_ - t0).count() : ms_since(t0);      // the stages are split where the update was enqueued
      } else {
        vio.retrieveFromVisualSparseMap(img, vm.pv_list_);                           // vio.cpp:1808 (lean: pg stays on the device)
        c = ms_since(t0);
        vio.computeJacobianAndUpdateEKF(img);                                        // vio.cpp:1810, updates *state = _state
      }
      const double d0 = ms_since(t0);
      vm.JoinMapUpdate();                                                            // (async_map_update_: the frame is over when BOTH halves are; billed to the map-update stage)
      const double d = ms_since(t0);
      livo2_state s_vio; state.to_abi(s_vio);
      std::memcpy(&states[(f * 2) * sizeof(livo2_state) / 8], &s_lio, sizeof(livo2_state));
      std::memcpy(&states[(f * 2 + 1) * sizeof(livo2_state) / 8], &s_vio, sizeof(livo2_state));
      counts[f * 2] = vm.effct_feat_num_; counts[f * 2 + 1] = vio.total_points;
      if (grow && f < scripts.size()) {                                              // the map maintenance that closes processFrame (scripted): the NEXT frame's sync carries it to the device
        if (!load_grow_script(dir, (int)f, scripts[f])) { std::fprintf(stderr, "live_chain: grow%zu_* inputs missing\n", f); return 2; }
        replay_grow_script(vio, scripts[f], g_points, g_feats);
      }
      for (VisualPoint *pt : sm.voxel_points) for (int k = 0; k < 3; k++) sub_pos.push_back(pt->pos_[k]);
      stage[f * 5] = a; stage[f * 5 + 1] = (b - a) + (d - d0); stage[f * 5 + 2] = c - b; stage[f * 5 + 3] = d0 - c;
      if (std::getenv("LIVO2_SHIM_PROF")) std::fprintf(stderr, "frame %zu: map update call %.3f ms + join %.3f ms, its kernels %.1f us\n", f, b - a, d - d0, vm.last_map_kernel_us_);
      if (f >= warm) { total += d + (grow ? stage[f * 5 + 4] : 0.0); timed++; sub_pts += (size_t)vio.total_points; eff += (size_t)vm.effct_feat_num_; }   // the first frames are warm-up: allocations, pinned buffers,
                                                                                                                        // and the first update of a freshly built tree (every root voxel is new to it)
    }
    auto wr = [&](const char *name, const void *p, size_t bytes) { std::ofstream o(dir + "/" + name, std::ios::binary); o.write((const char *)p, bytes); };
    wr("live_out.bin", stage.data(), stage.size() * 8); wr("live_states.bin", states.data(), states.size() * 8);
    wr("live_counts.bin", counts.data(), counts.size() * 4); wr("live_sub_pos.bin", sub_pos.data(), sub_pos.size() * 8);
    if (std::getenv("LIVO2_SHIM_PROF")) for (size_t f = 0; f < F; f++) std::fprintf(stderr, "frame %zu: %.3f %.3f %.3f %.3f ms (sync %.3f)\n", f, stage[f * 5], stage[f * 5 + 1], stage[f * 5 + 2], stage[f * 5 + 3], stage[f * 5 + 4]);
    // per-stage MEDIAN over the timed frames (a frame in which a pool of the device tree grows, or the first update of a freshly built tree lands, costs tens of
    // milliseconds once; the median is the steady-state frame)
    double s[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
      std::vector<double> v;
      for (size_t f = warm; f < F; f++) v.push_back(stage[f * 5 + k]);
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      s[k] = v.size() % 2 ? v[v.size() / 2] : 0.5 * (v[v.size() / 2 - 1] + v[v.size() / 2]);
    }
    const double T = timed ? (double)timed : 1.0;
    if (grow)
      std::printf("live_chain%s grow: %zu frames timed, %.3f ms per frame (syncFeatMap %.3f [incremental: %d delta syncs, %d full], StateEstimation %.3f, UpdateVoxelMapFromPosterior %.3f, "
                  "retrieveFromVisualSparseMap %.3f, computeJacobianAndUpdateEKF %.3f); visual map %zu points / %zu observations at the end; mean scan %.0f points, sub-map %.0f patches; medians over the timed frames\n",
                  lean ? " (lean)" : "", timed, s[0] + s[1] + s[2] + s[3] + s[4], s[4], vio.delta_syncs_, vio.full_syncs_, s[0], s[1], s[2], s[3], vio.mirroredPoints(), vio.mirroredObservations(),
                  (double)off / (double)F, (double)sub_pts / T);
    else
    std::printf("live_chain%s: %zu frames timed, %.3f ms per frame (StateEstimation %.3f, UpdateVoxelMapFromPosterior %.3f, retrieveFromVisualSparseMap %.3f, computeJacobianAndUpdateEKF %.3f); "
                "mean scan %.0f points, effct_feat_num_ %.0f, sub-map %.0f patches; syncFeatMap %.3f ms per frame (outside the stages); medians over the timed frames, mean frame %.3f ms\n",
                lean ? " (lean)" : "", timed, s[0] + s[1] + s[2] + s[3], s[0], s[1], s[2], s[3], (double)off / (double)F, (double)eff / T, (double)sub_pts / T, s[4], total / T);
    for (auto &kv : vio.feat_map) delete kv.second;
  } catch (const std::exception &e) { std::fprintf(stderr, "live_chain: %s\n", e.what()); return 1; }
  return 0;
}
