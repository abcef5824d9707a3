// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by, or executed from the product path.
//
// CPU restatement of the per-scan stage in front of the LiDAR update:
//   backward propagation / undistortion of every raw point     src/IMU_Processing.cpp:494-539   (ImuProcess::UndistortPcl, LIO branch)
//   Exp(ang_vel, dt)                                            include/utils/so3_math.h:24-43
//   Pose6D                                                      msg/Pose6D.msg, include/common_lib.h:225-242 (set_pose6d)
//   voxel-grid centroid filter of the undistorted cloud         src/LIVMapper.cpp:351-352        (pcl::VoxelGrid<PointType>, leaf = filter_size_surf)
//
// pcl::VoxelGrid is third-party code that is NOT under /root/reference (PCL >= 1.8, system-installed, unpinned — README.md:57-61); its
// published algorithm (pcl/filters/impl/voxel_grid.hpp, applyFilter) is restated here: float min/max of the cloud, inverse_leaf_size =
// 1/leaf in float, min_b = floor(min * inv), leaf index = sum_k (floor(p_k * inv_k) - min_b_k) * divb_mul_k, points sorted by leaf index,
// one float-accumulated centroid per occupied leaf, output in ascending leaf-index order.  PCL sorts with std::sort (not stable), so the
// summation order inside a leaf is unspecified there; this restatement keeps the input order.  PARITY UNPINNED at this boundary.
#pragma once
#include "orc_math.hpp"
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace orc {

struct Pose6D { double offset_time, acc[3], gyr[3], vel[3], pos[3], rot[9]; };   // msg/Pose6D.msg

inline M3 ExpAngVel(const V3 &ang_vel, double dt) {                                // include/utils/so3_math.h:24-43
  const double ang_vel_norm = norm(ang_vel);
  if (ang_vel_norm > 0.0000001) {
    const V3 r_axis = ang_vel / ang_vel_norm;
    const M3 K = skew(r_axis);
    const double r_ang = ang_vel_norm * dt;
    return M3::Identity() + K * std::sin(r_ang) + (K * (1.0 - std::cos(r_ang))) * K;      // (1-cos)*K*K groups as ((1-cos)*K)*K
  }
  return M3::Identity();
}

// src/IMU_Processing.cpp:494-539.  xyz: [n][3] float (PointType x,y,z), curvature: [n] float (ms from the scan start), sorted ascending
// like pcl_wait_proc after sort(time_list) (IMU_Processing.cpp:154-156).  In place, exactly as the reference's backward loop walks.
inline void undistort_points(float *xyz, const float *curvature, int n, const Pose6D *IMUpose, int n_poses, const M3 &rot_end, const V3 &pos_end,
                             const M3 &Lid_rot_to_IMU, const V3 &Lid_offset_to_IMU) {
  if (n < 1 || n_poses < 2) return;
  int it_pcl = n - 1;
  const M3 extR_Ri = Lid_rot_to_IMU.T() * rot_end.T();
  const V3 exrR_extT = Lid_rot_to_IMU.T() * Lid_offset_to_IMU;
  for (int it_kp = n_poses - 1; it_kp != 0; it_kp--) {
    const Pose6D &head = IMUpose[it_kp - 1];
    M3 R_imu; std::memcpy(R_imu.a, head.rot, 72);
    const V3 acc_imu = vec3(head.acc[0], head.acc[1], head.acc[2]), vel_imu = vec3(head.vel[0], head.vel[1], head.vel[2]);
    const V3 pos_imu = vec3(head.pos[0], head.pos[1], head.pos[2]), angvel_avr = vec3(head.gyr[0], head.gyr[1], head.gyr[2]);
    for (; curvature[it_pcl] / double(1000) > head.offset_time; it_pcl--) {
      const double dt = curvature[it_pcl] / double(1000) - head.offset_time;
      const M3 R_i = R_imu * ExpAngVel(angvel_avr, dt);
      const V3 T_ei = pos_imu + vel_imu * dt + ((acc_imu * 0.5) * dt) * dt - pos_end;      // 0.5 * acc_imu * dt * dt evaluates left to right: ((0.5*a)*dt)*dt
      const V3 P_i = vec3(xyz[3 * it_pcl], xyz[3 * it_pcl + 1], xyz[3 * it_pcl + 2]);
      const V3 P_compensate = extR_Ri * (R_i * (Lid_rot_to_IMU * P_i + Lid_offset_to_IMU) + T_ei) - exrR_extT;
      xyz[3 * it_pcl] = (float)P_compensate[0]; xyz[3 * it_pcl + 1] = (float)P_compensate[1]; xyz[3 * it_pcl + 2] = (float)P_compensate[2];
      if (it_pcl == 0) break;
    }
  }
}

// pcl::VoxelGrid::applyFilter restated (see the header comment).  Returns the number of output points, -1 if the leaf grid would overflow
// int32 (PCL prints a warning and returns the input cloud unchanged).
inline int voxel_grid_filter(const float *xyz, int n, float leaf, std::vector<float> &out) {
  out.clear();
  if (n == 0) return 0;
  float mn[3] = {xyz[0], xyz[1], xyz[2]}, mx[3] = {xyz[0], xyz[1], xyz[2]};
  for (int i = 1; i < n; i++) for (int k = 0; k < 3; k++) { mn[k] = std::min(mn[k], xyz[3 * i + k]); mx[k] = std::max(mx[k], xyz[3 * i + k]); }
  const float inv = 1.0f / leaf;
  int64_t min_b[3], max_b[3], div_b[3];
  for (int k = 0; k < 3; k++) { min_b[k] = (int64_t)std::floor(mn[k] * inv); max_b[k] = (int64_t)std::floor(mx[k] * inv); div_b[k] = max_b[k] - min_b[k] + 1; }
  if (div_b[0] * div_b[1] * div_b[2] > (int64_t)INT32_MAX) return -1;
  const int mul[3] = {1, (int)div_b[0], (int)(div_b[0] * div_b[1])};
  std::vector<std::pair<int, int>> index((size_t)n);      // (leaf index, point index)
  for (int i = 0; i < n; i++) {
    int idx = 0;
    for (int k = 0; k < 3; k++) idx += (int)(std::floor(xyz[3 * i + k] * inv) - (float)min_b[k]) * mul[k];
    index[i] = {idx, i};
  }
  std::stable_sort(index.begin(), index.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
  for (size_t first = 0; first < index.size();) {
    size_t last = first;
    float c[3] = {0.f, 0.f, 0.f};
    while (last < index.size() && index[last].first == index[first].first) {
      for (int k = 0; k < 3; k++) c[k] += xyz[3 * index[last].second + k];
      last++;
    }
    const float cnt = (float)(last - first);
    for (int k = 0; k < 3; k++) out.push_back(c[k] / cnt);
    first = last;
  }
  return (int)(out.size() / 3);
}

} // namespace orc
