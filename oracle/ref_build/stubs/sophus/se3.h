// TEST INFRASTRUCTURE ONLY — stand-in for <sophus/se3.h> (non-templated Sophus, commit a621ff per the reference's README.md:63-74, absent
// from this image).  A rigid transform p -> R p + t with the members the reference calls.  Real Sophus keeps the rotation as a unit
// quaternion; the algebra is the same rigid-transform algebra ("semantically unambiguous"), rounding differs at 1e-16.
#pragma once
#include <Eigen/Dense>
namespace Sophus {
class SE3 {
  Eigen::Matrix3d R_;
  Eigen::Vector3d t_;
public:
  SE3() : R_(Eigen::Matrix3d::Identity()), t_(Eigen::Vector3d::Zero()) {}
  SE3(const Eigen::Matrix3d &R, const Eigen::Vector3d &t) : R_(R), t_(t) {}
  Eigen::Matrix3d rotation_matrix() const { return R_; }
  const Eigen::Vector3d &translation() const { return t_; }
  Eigen::Vector3d &translation() { return t_; }
  SE3 inverse() const { Eigen::Matrix3d Rt = R_.transpose(); return SE3(Rt, -(Rt * t_)); }
  SE3 operator*(const SE3 &o) const { return SE3(R_ * o.R_, R_ * o.t_ + t_); }
  Eigen::Vector3d operator*(const Eigen::Vector3d &p) const { return R_ * p + t_; }
};
} // namespace Sophus
