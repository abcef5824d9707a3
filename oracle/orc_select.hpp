// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the selection half of VIOManager::retrieveFromVisualSparseMap (raycast_en = false as in
// config/avia.yaml:36):
//   A. scan points -> sub_feat_map voxels + depth image                      src/vio.cpp:385-427
//   B. visual points of those voxels -> nearest point per grid cell          src/vio.cpp:438-486
//   C. per selected cell: depth-continuity test against the depth image      src/vio.cpp:598-635
//   insertPointIntoVoxelMap's key (the key a visual point is FILED under)    src/vio.cpp:227-244
//   grid geometry / reset values                                             src/vio.cpp:67-78, 162-177
// Quirks restated as they are: the lookup key of a scan point is floor(p / 0.5) and then ANOTHER -1 for negatives (vio.cpp:392-396) while
// visual points are filed under (int64)(float(p / 0.5) - 1 for negatives) (vio.cpp:232-236), so along a negative axis a scan point looks
// into the voxel one below its own; the depth image keeps the LAST scan point that lands on a pixel; `cur_dist <= map_dist` lets the last
// visited of equally distant points win (visit order = unordered_map order in the reference, voxel insertion order here: exact float ties
// are the only place where that can matter).
// Third-party (rpg_vikit, unpinned — parity unpinned): world2cam (pinhole, zero distortion) and
//   AbstractCamera::isInFrame(Vector2i obs, int boundary) = obs.x >= boundary && obs.x < width - boundary && obs.y >= boundary && obs.y < height - boundary.
#pragma once
#include "orc_visual.hpp"
#include <map>
#include <unordered_map>

namespace orc {

struct SelectCfg {
  PinholeCam cam;
  M3 R_cur; V3 t_cur;                       // new_frame_->T_f_w_
  int border, grid_size, grid_n_width, grid_n_height, patch_size_half;
};

struct VisualMapPoint { V3 pos; int64_t key[3]; int active; };     // pos_, the feat_map voxel it is filed under, pt != nullptr && obs_.size() > 0

inline void feat_map_key(const V3 &pt_w, int64_t key[3]) {            // insertPointIntoVoxelMap, src/vio.cpp:227-236
  const double voxel_size = 0.5;
  for (int j = 0; j < 3; j++) {
    float loc = (float)(pt_w[j] / voxel_size);
    if (loc < 0) loc -= 1.0;
    key[j] = (int64_t)loc;
  }
}

inline bool isInFrame(const PinholeCam &c, int x, int y, int boundary) { return x >= boundary && x < c.width - boundary && y >= boundary && y < c.height - boundary; }

struct KeyLess { bool operator()(const std::array<int64_t, 3> &a, const std::array<int64_t, 3> &b) const { return a < b; } };

// cell_point[length]: index of the selected visual point or -1 (grid_num == TYPE_MAP <=> cell_type = 1); cell_dist = map_dist; discont[length]:
// 1 if the depth-continuity test rejects the selected point; in_fov[n_points]: the point passed isInFrame (voxel_in_fov = any of its points);
// depth_img: height x width float.
inline void visual_select(const SelectCfg &cfg, const double *pg, int n_pg, const VisualMapPoint *pts, int n_pts, int *cell_point, float *cell_dist, int *cell_type,
                          int *discont, int *in_fov, float *depth_img) {
  const int width = cfg.cam.width, height = cfg.cam.height, length = cfg.grid_n_width * cfg.grid_n_height;
  for (int i = 0; i < length; i++) { cell_point[i] = -1; cell_dist[i] = 10000.0f; cell_type[i] = 0; discont[i] = 0; }
  for (int i = 0; i < n_pts; i++) in_fov[i] = 0;
  for (size_t i = 0; i < (size_t)width * height; i++) depth_img[i] = 0.f;
  // A
  const float voxel_size = 0.5;
  std::map<std::array<int64_t, 3>, int, KeyLess> sub_feat_map;      // key -> insertion rank
  std::vector<std::array<int64_t, 3>> order;
  for (int i = 0; i < n_pg; i++) {
    const V3 pt_w = vec3(pg[3 * i], pg[3 * i + 1], pg[3 * i + 2]);
    int loc_xyz[3];
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = (int)std::floor(pt_w[j] / voxel_size);
      if (loc_xyz[j] < 0) loc_xyz[j] = (int)(loc_xyz[j] - 1.0);
    }
    const std::array<int64_t, 3> position = {loc_xyz[0], loc_xyz[1], loc_xyz[2]};
    if (sub_feat_map.emplace(position, (int)order.size()).second) order.push_back(position);
    const V3 pt_c = cfg.R_cur * pt_w + cfg.t_cur;
    if (pt_c[2] > 0) {
      double px[2];
      cfg.cam.world2cam(pt_c, px);
      if (isInFrame(cfg.cam, (int)px[0], (int)px[1], cfg.border)) {
        const float depth = (float)pt_c[2];
        const int col = (int)px[0], row = (int)px[1];
        depth_img[width * row + col] = depth;
      }
    }
  }
  // B: voxel -> its visual points (feat_map), visited in sub_feat_map order
  std::map<std::array<int64_t, 3>, std::vector<int>, KeyLess> feat_map;
  for (int i = 0; i < n_pts; i++) feat_map[{pts[i].key[0], pts[i].key[1], pts[i].key[2]}].push_back(i);
  const V3 cam_pos = (cfg.R_cur.T() * cfg.t_cur) * (-1.0);            // new_frame_->pos()
  for (const auto &position : order) {
    auto corre_voxel = feat_map.find(position);
    if (corre_voxel == feat_map.end()) continue;
    for (int i : corre_voxel->second) {
      if (!pts[i].active) continue;
      const V3 dir = cfg.R_cur * pts[i].pos + cfg.t_cur;
      if (dir[2] < 0) continue;
      double pc[2];
      cfg.cam.world2cam(dir, pc);
      if (isInFrame(cfg.cam, (int)pc[0], (int)pc[1], cfg.border)) {
        in_fov[i] = 1;
        const int index = (int)(pc[1] / cfg.grid_size) * cfg.grid_n_width + (int)(pc[0] / cfg.grid_size);
        cell_type[index] = 1;
        const V3 obs_vec = cam_pos - pts[i].pos;
        const float cur_dist = (float)norm(obs_vec);
        if (cur_dist <= cell_dist[index]) { cell_dist[index] = cur_dist; cell_point[index] = i; }
      }
    }
  }
  // C
  for (int i = 0; i < length; i++) {
    if (cell_type[i] != 1 || cell_point[i] < 0) continue;      // (a cell whose only points are farther than map_dist's initial 10000 keeps a null pointer in the reference)
    const VisualMapPoint &pt = pts[cell_point[i]];
    const V3 pt_cam = cfg.R_cur * pt.pos + cfg.t_cur;
    double pc[2];
    cfg.cam.world2cam(pt_cam, pc);
    bool depth_continous = false;
    for (int u = -cfg.patch_size_half; u <= cfg.patch_size_half && !depth_continous; u++)
      for (int v = -cfg.patch_size_half; v <= cfg.patch_size_half; v++) {
        if (u == 0 && v == 0) continue;
        const float depth = depth_img[width * (v + (int)pc[1]) + u + (int)pc[0]];
        if (depth == 0.) continue;
        const double delta_dist = std::fabs(pt_cam[2] - depth);
        if (delta_dist > 0.5) { depth_continous = true; break; }
      }
    discont[i] = depth_continous ? 1 : 0;
  }
}

} // namespace orc
