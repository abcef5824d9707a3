// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the LiDAR ESIKF measurement update of FAST-LIVO2:
//   calcBodyCov                         src/voxel_map.cpp:15-34
//   VoxelMapManager::StateEstimation    src/voxel_map.cpp:338-511
//   VoxelMapManager::TransformLidar     src/voxel_map.cpp:513-530   (float32 store, :524-526)
//   BuildResidualListOMP                src/voxel_map.cpp:643-711
//   build_single_residual               src/voxel_map.cpp:713-786
// including quirks Q1-Q10 of SURVEY.md §8a (float narrowing points, prior-vs-current pose mix, the unit-mismatched
// neighbour rule, all-8-children descent, DEG2RAD from PCL).  Data carriers are kept AoS like the reference
// (pointWithVar 384 B, PointToPlane ~488 B, global mutex around result writes) so that the timed build of this file is a
// representative CPU baseline ("port").
//
// PARITY STATUS: the reference cannot be compiled here (ROS/PCL/Eigen/OpenCV/Sophus/vikit absent) and ships no golden
// vectors for this path, so this oracle is "parity unpinned": it is pinned only by construction (line-by-line
// restatement), by hand-derived known-answer cases and by an independent numpy re-derivation (tests/).
#pragma once
#include "orc_state.hpp"
#include "orc_voxel_map.hpp"
#include <mutex>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

struct PointXYZINormal { float x, y, z, pad0, nx, ny, nz, pad1, intensity, curvature, pad2, pad3; };   // pcl::PointXYZINormal, 48 B
struct PointXYZI { float x, y, z, intensity; };

struct pointWithVar {                // include/common_lib.h:102-123
  V3 point_b, point_i, point_w;
  M3 var_nostate, body_var, var, point_crossmat;
  V3 normal;
  pointWithVar() {
    var_nostate = M3::Zero(); var = M3::Zero(); body_var = M3::Zero(); point_crossmat = M3::Zero();
    point_b = V3::Zero(); point_i = V3::Zero(); point_w = V3::Zero(); normal = V3::Zero();
  }
};

struct PointToPlane {                // include/voxel_map.h:54-67
  V3 point_b_, point_w_, normal_, center_;
  Mat<6, 6> plane_var_;
  M3 body_cov_;
  int layer_;
  double d_;
  double eigen_value_;
  bool is_valid_;
  float dis_to_plane_;
  const VoxelPlane *plane_src_;      // oracle-only: identity of the matched plane, for matched-set comparison
};

// src/voxel_map.cpp:15-34.  `deg2rad` stands for PCL's DEG2RAD factor (Q10; pcl_macros.h: ((x)*0.017453293)).
inline void calcBodyCov(V3 &pb, const float range_inc, const float degree_inc, M3 &cov, double deg2rad) {
  if (pb[2] == 0) pb[2] = 0.0001;
  float range = std::sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
  float range_var = range_inc * range_inc;
  double s = std::sin((degree_inc) * deg2rad);
  double dv = s * s;                               // pow(sin(.),2)
  Mat<2, 2> direction_var; direction_var(0, 0) = dv; direction_var(0, 1) = 0; direction_var(1, 0) = 0; direction_var(1, 1) = dv;
  V3 direction = pb;
  direction = direction / norm(direction);         // Eigen normalize(): *this /= norm()
  M3 direction_hat;
  direction_hat(0, 0) = 0; direction_hat(0, 1) = -direction[2]; direction_hat(0, 2) = direction[1];
  direction_hat(1, 0) = direction[2]; direction_hat(1, 1) = 0; direction_hat(1, 2) = -direction[0];
  direction_hat(2, 0) = -direction[1]; direction_hat(2, 1) = direction[0]; direction_hat(2, 2) = 0;
  V3 base_vector1 = vec3(1, 1, -(direction[0] + direction[1]) / direction[2]);
  base_vector1 = base_vector1 / norm(base_vector1);
  V3 base_vector2 = cross(base_vector1, direction);
  base_vector2 = base_vector2 / norm(base_vector2);
  Mat<3, 2> N;
  N(0, 0) = base_vector1[0]; N(0, 1) = base_vector2[0];
  N(1, 0) = base_vector1[1]; N(1, 1) = base_vector2[1];
  N(2, 0) = base_vector1[2]; N(2, 1) = base_vector2[2];
  Mat<3, 2> A = ((double)range * direction_hat) * N;
  cov = (direction * (double)range_var) * direction.T() + (A * direction_var) * A.T();
}

struct LidarIterTrace {              // what one ESIKF iteration produced (for per-iteration parity checks)
  int n_eff;
  double total_residual;
  double HtH[36], Htz[6];
  double solution[19];
  int converged, stopped;
};

class VoxelMapManager {              // include/voxel_map.h:187-256 (hot-path members only)
public:
  VoxelMapConfig config_setting_;
  VoxelMap *map_ = nullptr;          // owns voxel_map_ (reference member voxel_map_, voxel_map.h:194)
  std::vector<PointXYZINormal> feats_down_body_;
  M3 extR_; V3 extT_;
  StatesGroup state_;
  V3 position_last_;
  int feats_down_size_ = 0, effct_feat_num_ = 0;
  std::vector<M3> cross_mat_list_, body_cov_list_;
  std::vector<pointWithVar> pv_list_;
  std::vector<PointToPlane> ptpl_list_;
  std::vector<int> ptpl_index_;      // oracle-only: original point index of each ptpl_list_ entry
  double deg2rad_ = 0.017453293;
  int num_threads_ = 1;
  // oracle-only per-matched-point dumps of the LAST executed iteration
  std::vector<double> dump_Rinv_, dump_H_;
  std::vector<LidarIterTrace> trace_;

  // src/voxel_map.cpp:513-530
  void TransformLidar(const M3 &rot, const V3 &t, const std::vector<PointXYZINormal> &input_cloud, std::vector<PointXYZI> &trans_cloud) {
    std::vector<PointXYZI>().swap(trans_cloud);
    trans_cloud.reserve(input_cloud.size());
    for (size_t i = 0; i < input_cloud.size(); i++) {
      const PointXYZINormal &p_c = input_cloud[i];
      V3 p = vec3(p_c.x, p_c.y, p_c.z);
      p = (rot * (extR_ * p + extT_) + t);
      PointXYZI pi;
      pi.x = p[0]; pi.y = p[1]; pi.z = p[2];       // double -> float32 (Q1)
      pi.intensity = p_c.intensity;
      trans_cloud.push_back(pi);
    }
  }

  // src/voxel_map.cpp:713-786
  void build_single_residual(pointWithVar &pv, const VoxelOctoTree *current_octo, const int current_layer, bool &is_sucess, double &prob,
                             PointToPlane &single_ptpl) {
    int max_layer = config_setting_.max_layer_;
    double sigma_num = config_setting_.sigma_num_;
    double radius_k = 3;
    V3 p_w = pv.point_w;
    if (current_octo->plane_ptr_->is_plane_) {
      VoxelPlane &plane = *current_octo->plane_ptr_;
      float dis_to_plane = std::fabs(plane.normal_[0] * p_w[0] + plane.normal_[1] * p_w[1] + plane.normal_[2] * p_w[2] + plane.d_);
      float dis_to_center = (plane.center_[0] - p_w[0]) * (plane.center_[0] - p_w[0]) + (plane.center_[1] - p_w[1]) * (plane.center_[1] - p_w[1]) +
                            (plane.center_[2] - p_w[2]) * (plane.center_[2] - p_w[2]);
      float range_dis = std::sqrt(dis_to_center - dis_to_plane * dis_to_plane);   // float arithmetic, sqrtf (Q5); NaN when negative
      if (range_dis <= radius_k * plane.radius_) {
        Mat<1, 6> J_nq;
        for (int k = 0; k < 3; k++) { J_nq(0, k) = p_w[k] - plane.center_[k]; J_nq(0, 3 + k) = -plane.normal_[k]; }
        double sigma_l = ((J_nq * plane.plane_var_) * J_nq.T())(0, 0);
        sigma_l += ((plane.normal_.T() * pv.var) * plane.normal_)(0, 0);
        if (dis_to_plane < sigma_num * std::sqrt(sigma_l)) {
          is_sucess = true;
          double this_prob = 1.0 / (std::sqrt(sigma_l)) * std::exp(-0.5 * dis_to_plane * dis_to_plane / sigma_l);
          if (this_prob > prob) {
            prob = this_prob;
            pv.normal = plane.normal_;
            single_ptpl.body_cov_ = pv.body_var;
            single_ptpl.point_b_ = pv.point_b;
            single_ptpl.point_w_ = pv.point_w;
            single_ptpl.plane_var_ = plane.plane_var_;
            single_ptpl.normal_ = plane.normal_;
            single_ptpl.center_ = plane.center_;
            single_ptpl.d_ = plane.d_;
            single_ptpl.layer_ = current_layer;
            single_ptpl.dis_to_plane_ = plane.normal_[0] * p_w[0] + plane.normal_[1] * p_w[1] + plane.normal_[2] * p_w[2] + plane.d_;
            single_ptpl.plane_src_ = &plane;
          }
          return;
        } else { return; }
      } else { return; }
    } else {
      if (current_layer < max_layer) {
        for (size_t leafnum = 0; leafnum < 8; leafnum++) {
          if (current_octo->leaves_[leafnum] != nullptr) {
            VoxelOctoTree *leaf_octo = current_octo->leaves_[leafnum];
            build_single_residual(pv, leaf_octo, current_layer + 1, is_sucess, prob, single_ptpl);   // all 8 children (Q7)
          }
        }
        return;
      } else { return; }
    }
  }

  // src/voxel_map.cpp:643-711
  void BuildResidualListOMP(std::vector<pointWithVar> &pv_list, std::vector<PointToPlane> &ptpl_list) {
    double voxel_size = config_setting_.max_voxel_size_;
    std::mutex mylock;
    ptpl_list.clear();
    ptpl_index_.clear();
    std::vector<PointToPlane> all_ptpl_list(pv_list.size());
    std::vector<bool> useful_ptpl(pv_list.size());
    std::vector<size_t> index(pv_list.size());
    for (size_t i = 0; i < index.size(); ++i) { index[i] = i; useful_ptpl[i] = false; }
    auto &voxel_map_ = map_->voxel_map_;
#ifdef _OPENMP
    omp_set_num_threads(num_threads_);
#pragma omp parallel for
#endif
    for (int i = 0; i < (int)index.size(); i++) {
      pointWithVar &pv = pv_list[i];
      float loc_xyz[3];
      for (int j = 0; j < 3; j++) {
        loc_xyz[j] = pv.point_w[j] / voxel_size;       // double divide, narrowed to float (Q4)
        if (loc_xyz[j] < 0) { loc_xyz[j] -= 1.0; }
      }
      VOXEL_LOCATION position((int64_t)loc_xyz[0], (int64_t)loc_xyz[1], (int64_t)loc_xyz[2]);
      auto iter = voxel_map_.find(position);
      if (iter != voxel_map_.end()) {
        VoxelOctoTree *current_octo = iter->second;
        PointToPlane single_ptpl;
        bool is_sucess = false;
        double prob = 0;
        build_single_residual(pv, current_octo, 0, is_sucess, prob, single_ptpl);
        if (!is_sucess) {
          VOXEL_LOCATION near_position = position;     // voxel-index units compared with metres (Q3)
          if (loc_xyz[0] > (current_octo->voxel_center_[0] + current_octo->quater_length_)) { near_position.x = near_position.x + 1; }
          else if (loc_xyz[0] < (current_octo->voxel_center_[0] - current_octo->quater_length_)) { near_position.x = near_position.x - 1; }
          if (loc_xyz[1] > (current_octo->voxel_center_[1] + current_octo->quater_length_)) { near_position.y = near_position.y + 1; }
          else if (loc_xyz[1] < (current_octo->voxel_center_[1] - current_octo->quater_length_)) { near_position.y = near_position.y - 1; }
          if (loc_xyz[2] > (current_octo->voxel_center_[2] + current_octo->quater_length_)) { near_position.z = near_position.z + 1; }
          else if (loc_xyz[2] < (current_octo->voxel_center_[2] - current_octo->quater_length_)) { near_position.z = near_position.z - 1; }
          auto iter_near = voxel_map_.find(near_position);
          if (iter_near != voxel_map_.end()) { build_single_residual(pv, iter_near->second, 0, is_sucess, prob, single_ptpl); }
        }
        if (is_sucess) { mylock.lock(); useful_ptpl[i] = true; all_ptpl_list[i] = single_ptpl; mylock.unlock(); }
        else { mylock.lock(); useful_ptpl[i] = false; mylock.unlock(); }
      }
    }
    for (size_t i = 0; i < useful_ptpl.size(); i++) {
      if (useful_ptpl[i]) { ptpl_list.push_back(all_ptpl_list[i]); ptpl_index_.push_back((int)i); }
    }
  }

  // One pass of src/voxel_map.cpp:374-458 + 464-466: residual list, H / R^-1 / z assembly and the 6x6 / 6x1 reduction for the
  // CURRENT state_, without the solve.  Returns HtH (row-major 6x6) and Htz.
  void iterate_residual_and_reduce(const StatesGroup &state_propagat, double HtH6[36], double HTz6[6], double &total_residual) {
    total_residual = 0.0;
    std::vector<PointXYZI> world_lidar;
    TransformLidar(state_.rot_end, state_.pos_end, feats_down_body_, world_lidar);
    M3 rot_var, t_var;
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) { rot_var(r, c) = state_.cov(r, c); t_var(r, c) = state_.cov(3 + r, 3 + c); }
    for (size_t i = 0; i < feats_down_body_.size(); i++) {
      pointWithVar &pv = pv_list_[i];
      pv.point_b = vec3(feats_down_body_[i].x, feats_down_body_[i].y, feats_down_body_[i].z);
      pv.point_w = vec3(world_lidar[i].x, world_lidar[i].y, world_lidar[i].z);
      M3 cov = body_cov_list_[i];
      M3 point_crossmat = cross_mat_list_[i];
      cov = (state_.rot_end * cov) * state_.rot_end.T() + ((-point_crossmat) * rot_var) * (-(point_crossmat.T())) + t_var;   // :387 (no extR, Q6)
      pv.var = cov;
      pv.body_var = body_cov_list_[i];
    }
    ptpl_list_.clear();
    BuildResidualListOMP(pv_list_, ptpl_list_);
    for (size_t i = 0; i < ptpl_list_.size(); i++) total_residual += std::fabs(ptpl_list_[i].dis_to_plane_);
    effct_feat_num_ = (int)ptpl_list_.size();

    const int n = effct_feat_num_;
    std::vector<double> Hsub((size_t)n * 6), Hsub_T_R_inv((size_t)6 * n), R_inv(n), meas_vec(n, 0.0);
    for (int i = 0; i < n; i++) {
      auto &ptpl = ptpl_list_[i];
      V3 point_this = ptpl.point_b_;
      point_this = extR_ * point_this + extT_;
      M3 point_crossmat = skew(point_this);
      V3 point_world = state_propagat.rot_end * point_this + state_propagat.pos_end;       // PRIOR pose, un-rounded (Q1)
      Mat<1, 6> J_nq;
      for (int k = 0; k < 3; k++) { J_nq(0, k) = point_world[k] - ptpl.center_[k]; J_nq(0, 3 + k) = -ptpl.normal_[k]; }
      M3 RE = state_propagat.rot_end * extR_;
      M3 var = (RE * ptpl.body_cov_) * RE.T();                                             // :445 (prior rotation, Q2)
      double sigma_l = ((J_nq * ptpl.plane_var_) * J_nq.T())(0, 0);
      R_inv[i] = 1.0 / (0.001 + sigma_l + ((ptpl.normal_.T() * var) * ptpl.normal_)(0, 0));
      V3 A = (point_crossmat * state_.rot_end.T()) * ptpl.normal_;                          // CURRENT rotation (:453)
      double h[6] = {A[0], A[1], A[2], ptpl.normal_[0], ptpl.normal_[1], ptpl.normal_[2]};
      for (int k = 0; k < 6; k++) { Hsub[(size_t)i * 6 + k] = h[k]; Hsub_T_R_inv[(size_t)k * n + i] = h[k] * R_inv[i]; }
      meas_vec[i] = -ptpl.dis_to_plane_;
    }
    // HTz = Hsub_T_R_inv * meas_vec (:464, a column-major GEMV in Eigen), H_T_H = Hsub_T_R_inv * Hsub (:466, a GEMM); order of additions: orc_math.hpp
    for (int a = 0; a < 6; a++) {
      HTz6[a] = long_dot_gemv_colmajor(&Hsub_T_R_inv[(size_t)a * n], 1, meas_vec.data(), 1, (size_t)n);
      for (int b = 0; b < 6; b++) HtH6[a * 6 + b] = long_dot_gemm(&Hsub_T_R_inv[(size_t)a * n], 1, &Hsub[b], 6, (size_t)n);
    }
    dump_Rinv_ = R_inv; dump_H_ = Hsub;
  }

  // src/voxel_map.cpp:349-363: once-per-scan precompute
  void per_scan_precompute() {
    cross_mat_list_.clear(); cross_mat_list_.reserve(feats_down_size_);
    body_cov_list_.clear(); body_cov_list_.reserve(feats_down_size_);
    for (size_t i = 0; i < feats_down_body_.size(); i++) {
      V3 point_this = vec3(feats_down_body_[i].x, feats_down_body_[i].y, feats_down_body_[i].z);
      if (point_this[2] == 0) { point_this[2] = 0.001; }
      M3 var;
      calcBodyCov(point_this, config_setting_.dept_err_, config_setting_.beam_err_, var, deg2rad_);
      body_cov_list_.push_back(var);
      point_this = extR_ * point_this + extT_;
      cross_mat_list_.push_back(skew(point_this));
    }
    std::vector<pointWithVar>().swap(pv_list_);
    pv_list_.resize(feats_down_size_);
  }

  // src/voxel_map.cpp:338-511
  void StateEstimation(StatesGroup &state_propagat) {
    per_scan_precompute();
    trace_.clear();
    int rematch_num = 0;
    MState G = MState::Zero(), H_T_H = MState::Zero(), I_STATE = MState::Identity();
    bool flg_EKF_converged, EKF_stop_flg = 0;
    for (int iterCount = 0; iterCount < config_setting_.max_iterations_; iterCount++) {
      LidarIterTrace tr; std::memset(&tr, 0, sizeof(tr));
      double HtH6[36], HTz6[6], total_residual;
      iterate_residual_and_reduce(state_propagat, HtH6, HTz6, total_residual);
      tr.n_eff = effct_feat_num_; tr.total_residual = total_residual;
      std::memcpy(tr.HtH, HtH6, sizeof(HtH6)); std::memcpy(tr.Htz, HTz6, sizeof(HTz6));
      EKF_stop_flg = false;
      flg_EKF_converged = false;
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) H_T_H(a, b) = HtH6[a * 6 + b];
      MState Pinv, K_1;
      inverse_lu<ORC_DIM_STATE>(state_.cov, Pinv);
      inverse_lu<ORC_DIM_STATE>(H_T_H + Pinv, K_1);                                          // :468
      for (int r = 0; r < ORC_DIM_STATE; r++)
        for (int c = 0; c < 6; c++) { double s = K_1(r, 0) * H_T_H(0, c); for (int k = 1; k < 6; k++) s = s + K_1(r, k) * H_T_H(k, c); G(r, c) = s; }   // :469
      VState vec = state_propagat - state_;                                                  // :470
      VState solution;
      for (int r = 0; r < ORC_DIM_STATE; r++) {                                              // :472
        double kz = K_1(r, 0) * HTz6[0]; for (int k = 1; k < 6; k++) kz = kz + K_1(r, k) * HTz6[k];
        double gv = G(r, 0) * vec[0]; for (int k = 1; k < 6; k++) gv = gv + G(r, k) * vec[k];
        solution[r] = kz + vec[r] - gv;
      }
      state_ += solution;                                                                    // :474
      V3 rot_add = vec3(solution[0], solution[1], solution[2]);
      V3 t_add = vec3(solution[3], solution[4], solution[5]);
      if ((norm(rot_add) * 57.3 < 0.01) && (norm(t_add) * 100 < 0.015)) { flg_EKF_converged = true; }   // :477
      if (flg_EKF_converged || ((rematch_num == 0) && (iterCount == (config_setting_.max_iterations_ - 2)))) { rematch_num++; }   // :482
      if (!EKF_stop_flg && (rematch_num >= 2 || (iterCount == config_setting_.max_iterations_ - 1))) {   // :485
        state_.cov = (I_STATE - G) * state_.cov;                                             // :489-490
        position_last_ = state_.pos_end;
        EKF_stop_flg = true;
      }
      std::memcpy(tr.solution, solution.a, sizeof(tr.solution));
      tr.converged = flg_EKF_converged; tr.stopped = EKF_stop_flg;
      trace_.push_back(tr);
      if (EKF_stop_flg) break;
    }
  }
};

} // namespace orc
