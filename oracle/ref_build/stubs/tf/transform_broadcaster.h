// TEST INFRASTRUCTURE ONLY — stand-in for <tf/transform_broadcaster.h>.  createQuaternionMsgFromRollPitchYaw follows tf's documented
// fixed-axis roll/pitch/yaw composition (tf::Quaternion::setRPY); the reference calls it at voxel_map.cpp:493 for geoQuat_ only.
#pragma once
#include <cmath>
#include <geometry_msgs/Quaternion.h>
#include <omp.h>
#include <ros/ros.h>
namespace tf {
inline geometry_msgs::Quaternion createQuaternionMsgFromRollPitchYaw(double roll, double pitch, double yaw) {
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr), sr = std::sin(hr);
  geometry_msgs::Quaternion q;
  q.x = sr * cp * cy - cr * sp * sy;
  q.y = cr * sp * cy + sr * cp * sy;
  q.z = cr * cp * sy - sr * sp * cy;
  q.w = cr * cp * cy + sr * sp * sy;
  return q;
}
} // namespace tf
