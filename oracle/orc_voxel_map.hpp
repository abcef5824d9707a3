// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// Restatement of the reference VoxelMap data carriers and of the map construction that feeds the hot path:
//   VoxelPlane / VOXEL_LOCATION(+hash) / VoxelOctoTree   include/voxel_map.h:69-183
//   init_plane / init_octo_tree / cut_octo_tree / UpdateOctoTree   src/voxel_map.cpp:55-290
//   BuildVoxelMap / UpdateVoxelMap   src/voxel_map.cpp:532-591, 609-641
// Map construction is NOT on the hot path; it is restated here only so that the synthetic
// scenarios carry realistic plane_var_/radius_/octree shapes.  The 3x3 eigen-decomposition is a cyclic Jacobi
// solver (the reference uses Eigen::EigenSolver, voxel_map.cpp:70) — eigenvector signs/ordering of ties differ,
// which is irrelevant for a scenario generator.
#pragma once
#include "orc_math.hpp"
#include <unordered_map>
#include <vector>

namespace orc {

#define ORC_HASH_P 116101            // include/voxel_map.h:30
#define ORC_MAX_N 10000000000LL      // include/voxel_map.h:31

struct VOXEL_LOCATION {              // include/voxel_map.h:96-104
  int64_t x, y, z;
  VOXEL_LOCATION(int64_t vx = 0, int64_t vy = 0, int64_t vz = 0) : x(vx), y(vy), z(vz) {}
  bool operator==(const VOXEL_LOCATION &o) const { return (x == o.x && y == o.y && z == o.z); }
};
struct VoxelHash {                   // include/voxel_map.h:109-117
  size_t operator()(const VOXEL_LOCATION &s) const {
    return (size_t)(((((s.z) * ORC_HASH_P) % ORC_MAX_N + (s.y)) * ORC_HASH_P) % ORC_MAX_N + (s.x));
  }
};

struct MapPoint { V3 point_w; M3 var; };   // the subset of pointWithVar the map builder reads (common_lib.h:102-123)

struct VoxelPlane {                  // include/voxel_map.h:69-94 (fields the path or the builder touches)
  V3 center_, normal_, y_normal_, x_normal_;
  M3 covariance_;
  Mat<6, 6> plane_var_;
  float radius_ = 0, min_eigen_value_ = 1, mid_eigen_value_ = 1, max_eigen_value_ = 1, d_ = 0;
  int points_size_ = 0;
  bool is_plane_ = false, is_init_ = false, is_update_ = false;
  int id_ = 0;
  VoxelPlane() { plane_var_ = Mat<6, 6>::Zero(); covariance_ = M3::Zero(); center_ = V3::Zero(); normal_ = V3::Zero(); y_normal_ = V3::Zero(); x_normal_ = V3::Zero(); }
};

// symmetric 3x3 eigen-decomposition (cyclic Jacobi); columns of V are unit eigenvectors
inline void eig3_sym(const M3 &A, double ev[3], M3 &V) {
  M3 a = A; V = M3::Identity();
  for (int sweep = 0; sweep < 64; sweep++) {
    double off = a(0, 1) * a(0, 1) + a(0, 2) * a(0, 2) + a(1, 2) * a(1, 2);
    if (off < 1e-300) break;
    for (int p = 0; p < 2; p++)
      for (int q = p + 1; q < 3; q++) {
        if (a(p, q) == 0.0) continue;
        double theta = (a(q, q) - a(p, p)) / (2.0 * a(p, q));
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; k++) { double akp = a(k, p), akq = a(k, q); a(k, p) = c * akp - s * akq; a(k, q) = s * akp + c * akq; }
        for (int k = 0; k < 3; k++) { double apk = a(p, k), aqk = a(q, k); a(p, k) = c * apk - s * aqk; a(q, k) = s * apk + c * aqk; }
        for (int k = 0; k < 3; k++) { double vkp = V(k, p), vkq = V(k, q); V(k, p) = c * vkp - s * vkq; V(k, q) = s * vkp + c * vkq; }
      }
  }
  ev[0] = a(0, 0); ev[1] = a(1, 1); ev[2] = a(2, 2);
}

struct VoxelOctoTree {               // include/voxel_map.h:129-183
  std::vector<MapPoint> temp_points_;
  VoxelPlane *plane_ptr_;
  int layer_;
  int octo_state_;
  VoxelOctoTree *leaves_[8];
  double voxel_center_[3];
  std::vector<int> layer_init_num_;
  float quater_length_;
  float planer_threshold_;
  int points_size_threshold_, update_size_threshold_, max_points_num_, max_layer_, new_points_;
  bool init_octo_, update_enable_;
  int *plane_id_counter_;

  VoxelOctoTree(int max_layer, int layer, int points_size_threshold, int max_points_num, float planer_threshold, int *id_counter)
      : layer_(layer), planer_threshold_(planer_threshold), points_size_threshold_(points_size_threshold), max_points_num_(max_points_num),
        max_layer_(max_layer), plane_id_counter_(id_counter) {
    octo_state_ = 0; new_points_ = 0; update_size_threshold_ = 5; init_octo_ = false; update_enable_ = true;
    for (int i = 0; i < 8; i++) leaves_[i] = nullptr;
    plane_ptr_ = new VoxelPlane;
    voxel_center_[0] = voxel_center_[1] = voxel_center_[2] = 0; quater_length_ = 0;
  }
  ~VoxelOctoTree() { for (int i = 0; i < 8; i++) delete leaves_[i]; delete plane_ptr_; }

  // src/voxel_map.cpp:292-305
  VoxelOctoTree *find_correspond(const V3 &pw) {
    if (!init_octo_ || plane_ptr_->is_plane_ || (layer_ >= max_layer_)) return this;
    int xyz[3] = {0, 0, 0};
    xyz[0] = pw[0] > voxel_center_[0] ? 1 : 0;
    xyz[1] = pw[1] > voxel_center_[1] ? 1 : 0;
    xyz[2] = pw[2] > voxel_center_[2] ? 1 : 0;
    int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
    return (leaves_[leafnum] != nullptr) ? leaves_[leafnum]->find_correspond(pw) : this;
  }

  // src/voxel_map.cpp:55-135
  void init_plane(const std::vector<MapPoint> &points, VoxelPlane *plane) {
    plane->plane_var_ = Mat<6, 6>::Zero();
    plane->covariance_ = M3::Zero();
    plane->center_ = V3::Zero();
    plane->normal_ = V3::Zero();
    plane->points_size_ = (int)points.size();
    plane->radius_ = 0;
    for (const auto &pv : points) {
      plane->covariance_ = plane->covariance_ + pv.point_w * pv.point_w.T();
      plane->center_ = plane->center_ + pv.point_w;
    }
    plane->center_ = plane->center_ / (double)plane->points_size_;
    plane->covariance_ = plane->covariance_ / (double)plane->points_size_ - plane->center_ * plane->center_.T();
    double evalsReal[3]; M3 evecs;
    eig3_sym(plane->covariance_, evalsReal, evecs);
    int evalsMin = 0, evalsMax = 0;
    for (int i = 1; i < 3; i++) { if (evalsReal[i] < evalsReal[evalsMin]) evalsMin = i; if (evalsReal[i] > evalsReal[evalsMax]) evalsMax = i; }
    if (evalsMin == evalsMax) { evalsMin = 0; evalsMax = 2; }
    int evalsMid = 3 - evalsMin - evalsMax;
    M3 J_Q = M3::Zero();
    J_Q(0, 0) = J_Q(1, 1) = J_Q(2, 2) = 1.0 / plane->points_size_;
    if (evalsReal[evalsMin] < planer_threshold_) {
      V3 vmin = vec3(evecs(0, evalsMin), evecs(1, evalsMin), evecs(2, evalsMin));
      for (size_t i = 0; i < points.size(); i++) {
        Mat<6, 3> J; M3 F;
        for (int m = 0; m < 3; m++) {
          if (m != evalsMin) {
            V3 vm = vec3(evecs(0, m), evecs(1, m), evecs(2, m));
            Mat<1, 3> lhs = (points[i].point_w - plane->center_).T() / ((plane->points_size_) * (evalsReal[evalsMin] - evalsReal[m]));
            Mat<1, 3> F_m = lhs * (vm * vmin.T() + vmin * vm.T());
            for (int c = 0; c < 3; c++) F(m, c) = F_m(0, c);
          } else { for (int c = 0; c < 3; c++) F(m, c) = 0; }
        }
        M3 top = evecs * F;
        for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { J(r, c) = top(r, c); J(3 + r, c) = J_Q(r, c); }
        plane->plane_var_ = plane->plane_var_ + J * points[i].var * J.T();
      }
      plane->normal_ = vmin;
      plane->y_normal_ = vec3(evecs(0, evalsMid), evecs(1, evalsMid), evecs(2, evalsMid));
      plane->x_normal_ = vec3(evecs(0, evalsMax), evecs(1, evalsMax), evecs(2, evalsMax));
      plane->min_eigen_value_ = evalsReal[evalsMin];
      plane->mid_eigen_value_ = evalsReal[evalsMid];
      plane->max_eigen_value_ = evalsReal[evalsMax];
      plane->radius_ = std::sqrt(evalsReal[evalsMax]);
      plane->d_ = -(plane->normal_[0] * plane->center_[0] + plane->normal_[1] * plane->center_[1] + plane->normal_[2] * plane->center_[2]);
      plane->is_plane_ = true;
      plane->is_update_ = true;
      if (!plane->is_init_) { plane->id_ = *plane_id_counter_; (*plane_id_counter_)++; plane->is_init_ = true; }
    } else {
      plane->is_update_ = true;
      plane->is_plane_ = false;
    }
  }

  VoxelOctoTree *new_leaf(const int xyz[3]) {
    VoxelOctoTree *l = new VoxelOctoTree(max_layer_, layer_ + 1, layer_init_num_[layer_ + 1], max_points_num_, planer_threshold_, plane_id_counter_);
    l->layer_init_num_ = layer_init_num_;
    l->voxel_center_[0] = voxel_center_[0] + (2 * xyz[0] - 1) * quater_length_;
    l->voxel_center_[1] = voxel_center_[1] + (2 * xyz[1] - 1) * quater_length_;
    l->voxel_center_[2] = voxel_center_[2] + (2 * xyz[2] - 1) * quater_length_;
    l->quater_length_ = quater_length_ / 2;
    return l;
  }

  // src/voxel_map.cpp:137-161
  void init_octo_tree() {
    if ((int)temp_points_.size() > points_size_threshold_) {
      init_plane(temp_points_, plane_ptr_);
      if (plane_ptr_->is_plane_ == true) {
        octo_state_ = 0;
        if ((int)temp_points_.size() > max_points_num_) { update_enable_ = false; std::vector<MapPoint>().swap(temp_points_); new_points_ = 0; }
      } else { octo_state_ = 1; cut_octo_tree(); }
      init_octo_ = true;
      new_points_ = 0;
    }
  }

  // src/voxel_map.cpp:163-217
  void cut_octo_tree() {
    if (layer_ >= max_layer_) { octo_state_ = 0; return; }
    for (size_t i = 0; i < temp_points_.size(); i++) {
      int xyz[3] = {0, 0, 0};
      if (temp_points_[i].point_w[0] > voxel_center_[0]) xyz[0] = 1;
      if (temp_points_[i].point_w[1] > voxel_center_[1]) xyz[1] = 1;
      if (temp_points_[i].point_w[2] > voxel_center_[2]) xyz[2] = 1;
      int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
      if (leaves_[leafnum] == nullptr) leaves_[leafnum] = new_leaf(xyz);
      leaves_[leafnum]->temp_points_.push_back(temp_points_[i]);
      leaves_[leafnum]->new_points_++;
    }
    for (unsigned i = 0; i < 8; i++) {
      if (leaves_[i] != nullptr) {
        if ((int)leaves_[i]->temp_points_.size() > leaves_[i]->points_size_threshold_) {
          init_plane(leaves_[i]->temp_points_, leaves_[i]->plane_ptr_);
          if (leaves_[i]->plane_ptr_->is_plane_) {
            leaves_[i]->octo_state_ = 0;
            if ((int)leaves_[i]->temp_points_.size() > leaves_[i]->max_points_num_) {
              leaves_[i]->update_enable_ = false; std::vector<MapPoint>().swap(leaves_[i]->temp_points_); new_points_ = 0;
            }
          } else { leaves_[i]->octo_state_ = 1; leaves_[i]->cut_octo_tree(); }
          leaves_[i]->init_octo_ = true;
          leaves_[i]->new_points_ = 0;
        }
      }
    }
  }

  // src/voxel_map.cpp:219-290
  void UpdateOctoTree(const MapPoint &pv) {
    if (!init_octo_) {
      new_points_++; temp_points_.push_back(pv);
      if ((int)temp_points_.size() > points_size_threshold_) init_octo_tree();
    } else {
      if (plane_ptr_->is_plane_) {
        if (update_enable_) {
          new_points_++; temp_points_.push_back(pv);
          if (new_points_ > update_size_threshold_) { init_plane(temp_points_, plane_ptr_); new_points_ = 0; }
          if ((int)temp_points_.size() >= max_points_num_) { update_enable_ = false; std::vector<MapPoint>().swap(temp_points_); new_points_ = 0; }
        }
      } else {
        if (layer_ < max_layer_) {
          int xyz[3] = {0, 0, 0};
          if (pv.point_w[0] > voxel_center_[0]) xyz[0] = 1;
          if (pv.point_w[1] > voxel_center_[1]) xyz[1] = 1;
          if (pv.point_w[2] > voxel_center_[2]) xyz[2] = 1;
          int leafnum = 4 * xyz[0] + 2 * xyz[1] + xyz[2];
          if (leaves_[leafnum] != nullptr) leaves_[leafnum]->UpdateOctoTree(pv);
          else { leaves_[leafnum] = new_leaf(xyz); leaves_[leafnum]->UpdateOctoTree(pv); }
        } else {
          if (update_enable_) {
            new_points_++; temp_points_.push_back(pv);
            if (new_points_ > update_size_threshold_) { init_plane(temp_points_, plane_ptr_); new_points_ = 0; }
            if ((int)temp_points_.size() > max_points_num_) { update_enable_ = false; std::vector<MapPoint>().swap(temp_points_); new_points_ = 0; }
          }
        }
      }
    }
  }
};

struct VoxelMapConfig {              // include/voxel_map.h:35-52 (hot-path + builder knobs)
  double max_voxel_size_ = 0.5;
  int max_layer_ = 2;
  int max_iterations_ = 5;
  std::vector<int> layer_init_num_ = {5, 5, 5, 5, 5};
  int max_points_num_ = 50;
  double planner_threshold_ = 0.0025;
  double beam_err_ = 0.05, dept_err_ = 0.02, sigma_num_ = 3;
};

struct VoxelMap {
  VoxelMapConfig config_setting_;
  std::unordered_map<VOXEL_LOCATION, VoxelOctoTree *, VoxelHash> voxel_map_;
  int voxel_plane_id = 0;
  ~VoxelMap() { for (auto &kv : voxel_map_) delete kv.second; }

  static inline void voxel_key(const V3 &p_w, float voxel_size_f, int64_t key[3]) {
    // src/voxel_map.cpp:561-567 / 620-626: here voxel_size is a FLOAT local (534, 611), unlike BuildResidualListOMP (646)
    for (int j = 0; j < 3; j++) {
      float loc = p_w[j] / voxel_size_f;
      if (loc < 0) loc -= 1.0;
      key[j] = (int64_t)loc;
    }
  }
  VoxelOctoTree *new_root(const VOXEL_LOCATION &position, float voxel_size) {
    VoxelOctoTree *t = new VoxelOctoTree(config_setting_.max_layer_, 0, config_setting_.layer_init_num_[0], config_setting_.max_points_num_,
                                         (float)config_setting_.planner_threshold_, &voxel_plane_id);
    t->quater_length_ = voxel_size / 4;
    t->voxel_center_[0] = (0.5 + position.x) * voxel_size;
    t->voxel_center_[1] = (0.5 + position.y) * voxel_size;
    t->voxel_center_[2] = (0.5 + position.z) * voxel_size;
    t->layer_init_num_ = config_setting_.layer_init_num_;
    return t;
  }
  // src/voxel_map.cpp:532-591 (from the point where input_points exist)
  void BuildVoxelMap(const std::vector<MapPoint> &input_points) {
    float voxel_size = config_setting_.max_voxel_size_;
    for (size_t i = 0; i < input_points.size(); i++) {
      const MapPoint &p_v = input_points[i];
      int64_t k[3]; voxel_key(p_v.point_w, voxel_size, k);
      VOXEL_LOCATION position(k[0], k[1], k[2]);
      auto iter = voxel_map_.find(position);
      if (iter != voxel_map_.end()) { iter->second->temp_points_.push_back(p_v); iter->second->new_points_++; }
      else { VoxelOctoTree *t = new_root(position, voxel_size); voxel_map_[position] = t; t->temp_points_.push_back(p_v); t->new_points_++; }
    }
    for (auto iter = voxel_map_.begin(); iter != voxel_map_.end(); ++iter) iter->second->init_octo_tree();
  }
  // src/voxel_map.cpp:924-948 (mapSliding) + 950-972 (clearMemOutOfMap).  Returns the number of root voxels deleted, -1 when the threshold is not reached.
  V3 last_slide_position = vec3(0.0, 0.0, 0.0);                 // include/voxel_map.h:209
  int mapSliding(const V3 &position_last_, double sliding_thresh, int half_map_size) {
    if (norm(position_last_ - last_slide_position) < sliding_thresh) return -1;
    last_slide_position = position_last_;
    float loc_xyz[3];
    for (int j = 0; j < 3; j++) {
      loc_xyz[j] = position_last_[j] / config_setting_.max_voxel_size_;
      if (loc_xyz[j] < 0) loc_xyz[j] -= 1.0;
    }
    // clearMemOutOfMap(const int &x_max, ...): the int64 sums are narrowed to int at the call
    const int x_max = (int)((int64_t)loc_xyz[0] + half_map_size), x_min = (int)((int64_t)loc_xyz[0] - half_map_size);
    const int y_max = (int)((int64_t)loc_xyz[1] + half_map_size), y_min = (int)((int64_t)loc_xyz[1] - half_map_size);
    const int z_max = (int)((int64_t)loc_xyz[2] + half_map_size), z_min = (int)((int64_t)loc_xyz[2] - half_map_size);
    int delete_voxel_cout = 0;
    for (auto it = voxel_map_.begin(); it != voxel_map_.end();) {
      const VOXEL_LOCATION &loc = it->first;
      const bool should_remove = loc.x > x_max || loc.x < x_min || loc.y > y_max || loc.y < y_min || loc.z > z_max || loc.z < z_min;
      if (should_remove) { delete it->second; it = voxel_map_.erase(it); delete_voxel_cout++; }
      else ++it;
    }
    return delete_voxel_cout;
  }
  // src/voxel_map.cpp:609-641
  void UpdateVoxelMap(const std::vector<MapPoint> &input_points) {
    float voxel_size = config_setting_.max_voxel_size_;
    for (size_t i = 0; i < input_points.size(); i++) {
      const MapPoint &p_v = input_points[i];
      int64_t k[3]; voxel_key(p_v.point_w, voxel_size, k);
      VOXEL_LOCATION position(k[0], k[1], k[2]);
      auto iter = voxel_map_.find(position);
      if (iter != voxel_map_.end()) iter->second->UpdateOctoTree(p_v);
      else { VoxelOctoTree *t = new_root(position, voxel_size); voxel_map_[position] = t; t->UpdateOctoTree(p_v); }
    }
  }
};

} // namespace orc
