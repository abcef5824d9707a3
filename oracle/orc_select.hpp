// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the selection half of VIOManager::retrieveFromVisualSparseMap (raycast_en = false as in
// config/avia.yaml:36, and since round 4 the RayCasting module of raycast_en = true, src/vio.cpp:80-118 + 487-591):
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
#include "orc_warp.hpp"
#include <functional>
#include <map>
#include <unordered_map>

namespace orc {

struct SelectCfg {
  PinholeCam cam;
  M3 R_cur; V3 t_cur;                       // new_frame_->T_f_w_
  int border, grid_size, grid_n_width, grid_n_height, patch_size_half;
  int raycast_en = 0;                       // vio/raycast_en (LIVMapper.cpp:63, 141)
};

// plane_map.find(sample_pos) + find_correspond(sample_point_w) + plane_ptr_->is_plane_ (vio.cpp:573-585): true and (center_, normal_) when the ray stops at a plane;
// *root_found tells whether the voxel exists at all (a voxel without a plane lets the ray go on)
using PlaneLookup = std::function<bool(const int64_t key[3], const V3 &sample_point_w, V3 &center, V3 &normal)>;
struct RayHit { V3 center, normal; };

// rays_with_sample_points / border_flag of initializeVIO (vio.cpp:80-118): one ray per grid cell through the cell centre, sampled at depths 0.1, 0.3, ... (a FLOAT
// accumulator: d_temp += 0.2f while d_temp <= 3.0f), each sample = cam2world(u, v) scaled to z = d_temp
inline void raycast_rays(const SelectCfg &cfg, std::vector<std::vector<V3>> &rays, std::vector<int> &border_flag) {
  const int length = cfg.grid_n_width * cfg.grid_n_height;
  border_flag.assign(length, 0);
  rays.clear(); rays.reserve(length);
  float d_min = 0.1, d_max = 3.0, step = 0.2;
  for (int grid_row = 1; grid_row <= cfg.grid_n_height; grid_row++)
    for (int grid_col = 1; grid_col <= cfg.grid_n_width; grid_col++) {
      std::vector<V3> SamplePointsEachGrid;
      int index = (grid_row - 1) * cfg.grid_n_width + grid_col - 1;
      if (grid_row == 1 || grid_col == 1 || grid_row == cfg.grid_n_height || grid_col == cfg.grid_n_width) border_flag[index] = 1;
      int u = cfg.grid_size / 2 + (grid_col - 1) * cfg.grid_size;
      int v = cfg.grid_size / 2 + (grid_row - 1) * cfg.grid_size;
      for (float d_temp = d_min; d_temp <= d_max; d_temp += step) {
        V3 xyz = cam2world(cfg.cam, (double)u, (double)v);
        xyz = xyz * ((double)d_temp / xyz[2]);
        SamplePointsEachGrid.push_back(xyz);
      }
      rays.push_back(SamplePointsEachGrid);
    }
}

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
                          int *discont, int *in_fov, float *depth_img, const PlaneLookup *plane_lookup = nullptr, std::vector<RayHit> *add_from_voxel_map = nullptr) {
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
  // RayCasting module (vio.cpp:487-591): grid cells without a map point shoot a ray through their centre; the first sample whose voxel is already in sub_feat_map
  // ends the ray, the first that hits a feat_map voxel processes that voxel like stage B (and files it in sub_feat_map if something of it is in view), the first
  // that hits a LiDAR-map voxel whose leaf at the sample is a plane contributes (center_, normal_) to add_from_voxel_map.  The loop is ORDER-DEPENDENT as written in
  // the reference: a ray can turn a later cell into TYPE_MAP (that cell then shoots no ray) and fills sub_feat_map for the rays behind it.
  if (cfg.raycast_en) {
    std::vector<std::vector<V3>> rays; std::vector<int> border_flag;
    raycast_rays(cfg, rays, border_flag);
    const M3 Rt = cfg.R_cur.T();
    const V3 tinv = (Rt * cfg.t_cur) * (-1.0);                          // T_f_w_.inverse()
    for (int i = 0; i < length; i++) {
      if (cell_type[i] == 1 || border_flag[i] == 1) continue;
      for (const V3 &it : rays[i]) {
        const V3 sample_point_w = Rt * it + tinv;                       // new_frame_->f2w(it)
        int loc_xyz[3];
        for (int j = 0; j < 3; j++) {
          loc_xyz[j] = (int)std::floor(sample_point_w[j] / voxel_size);
          if (loc_xyz[j] < 0) loc_xyz[j] = (int)(loc_xyz[j] - 1.0);
        }
        const std::array<int64_t, 3> sample_pos = {loc_xyz[0], loc_xyz[1], loc_xyz[2]};
        if (sub_feat_map.find(sample_pos) != sub_feat_map.end()) break;
        auto corre_feat_map = feat_map.find(sample_pos);
        if (corre_feat_map != feat_map.end()) {
          bool voxel_in_fov = false;
          for (int k : corre_feat_map->second) {
            if (!pts[k].active) continue;
            const V3 dir = cfg.R_cur * pts[k].pos + cfg.t_cur;
            if (dir[2] < 0) continue;
            double pc[2];
            cfg.cam.world2cam(dir, pc);
            if (isInFrame(cfg.cam, (int)pc[0], (int)pc[1], cfg.border)) {
              voxel_in_fov = true;
              in_fov[k] = 1;
              const int index = (int)(pc[1] / cfg.grid_size) * cfg.grid_n_width + (int)(pc[0] / cfg.grid_size);
              cell_type[index] = 1;
              const V3 obs_vec = cam_pos - pts[k].pos;
              const float cur_dist = (float)norm(obs_vec);
              if (cur_dist <= cell_dist[index]) { cell_dist[index] = cur_dist; cell_point[index] = k; }
            }
          }
          if (voxel_in_fov) sub_feat_map.emplace(sample_pos, (int)order.size());
          break;
        } else if (plane_lookup) {
          const int64_t key[3] = {sample_pos[0], sample_pos[1], sample_pos[2]};
          RayHit h;
          if ((*plane_lookup)(key, sample_point_w, h.center, h.normal)) { if (add_from_voxel_map) add_from_voxel_map->push_back(h); break; }
        }
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
