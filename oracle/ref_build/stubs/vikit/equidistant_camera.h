// TEST INFRASTRUCTURE ONLY — stand-in for rpg_vikit's EquidistantCamera (see abstract_camera.h).  THIRD-PARTY, UNPINNED.
// world2cam: theta = atan(r), theta_d = theta (1 + k1 theta^2 + k2 theta^4 + k3 theta^6 + k4 theta^8), pixel = f * (x, y) * theta_d / r + c — the library's published
// Kannala-Brandt / OpenCV-fisheye form.  cam2world: the OpenCV-fisheye undistortPoints scheme (ten fixed-point iterations on theta), then normalisation.
#pragma once
#include <cmath>
#include <vikit/abstract_camera.h>
namespace vk {
class EquidistantCamera : public AbstractCamera {
  double fx_, fy_, cx_, cy_, k_[4], scale_;
public:
  EquidistantCamera(double width, double height, double scale, double fx, double fy, double cx, double cy, double k1, double k2, double k3, double k4)
      : AbstractCamera((int)(width * scale), (int)(height * scale)), fx_(fx * scale), fy_(fy * scale), cx_(cx * scale), cy_(cy * scale), k_{k1, k2, k3, k4}, scale_(scale) {}
  Vector3d cam2world(const double &u, const double &v) const override {
    const double xd = (u - cx_) / fx_, yd = (v - cy_) / fy_;
    const double thetad = std::sqrt(xd * xd + yd * yd);
    double theta = thetad;
    for (int j = 0; j < 10; j++) {
      const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
      theta = thetad / (1 + k_[0] * t2 + k_[1] * t4 + k_[2] * t6 + k_[3] * t8);
    }
    const double scaling = (thetad > 1e-8) ? std::tan(theta) / thetad : 1.0;
    Vector3d xyz(xd * scaling, yd * scaling, 1.0);
    return xyz.normalized();
  }
  Vector3d cam2world(const Vector2d &px) const override { return cam2world(px[0], px[1]); }
  Vector2d world2cam(const Vector3d &xyz_c) const override { return world2cam(Vector2d(xyz_c[0] / xyz_c[2], xyz_c[1] / xyz_c[2])); }
  Vector2d world2cam(const Vector2d &uv) const override {
    const double r = std::sqrt(uv[0] * uv[0] + uv[1] * uv[1]), theta = std::atan(r), t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
    const double thetad = theta * (1 + k_[0] * t2 + k_[1] * t4 + k_[2] * t6 + k_[3] * t8);
    const double scaling = (r > 1e-8) ? thetad / r : 1.0;
    Vector2d px;
    px[0] = fx_ * uv[0] * scaling + cx_; px[1] = fy_ * uv[1] * scaling + cy_;
    return px;
  }
  double errorMultiplier2() const override { return std::fabs(fx_); }
  double errorMultiplier() const override { return std::fabs(4.0 * fx_ * fy_); }
  double fx() const override { return fx_; }
  double fy() const override { return fy_; }
  double cx() const override { return cx_; }
  double cy() const override { return cy_; }
  double scale() const override { return scale_; }
};
} // namespace vk
