// TEST INFRASTRUCTURE ONLY — stand-in for <vikit/pinhole_camera.h> (see abstract_camera.h).  THIRD-PARTY, UNPINNED.
// world2cam: projection onto z = 1, optional radial-tangential distortion (d0, d1 radial, d2, d3 tangential, d4 = r^6), then fx, fy, cx, cy.
// cam2world: without distortion normalize((u-cx)/fx, (v-cy)/fy, 1); with distortion the library calls cv::undistortPoints on a CV_32FC2
// point — OpenCV's published five fixed-point iterations, float32 in and out.
#pragma once
#include <cmath>
#include <vikit/abstract_camera.h>
namespace vk {
class PinholeCamera : public AbstractCamera {
  double fx_, fy_, cx_, cy_, d_[5], scale_;
  bool distortion_;
public:
  PinholeCamera(double width, double height, double scale, double fx, double fy, double cx, double cy, double d0 = 0, double d1 = 0, double d2 = 0, double d3 = 0, double d4 = 0)
      : AbstractCamera((int)(width * scale), (int)(height * scale)), fx_(fx * scale), fy_(fy * scale), cx_(cx * scale), cy_(cy * scale), d_{d0, d1, d2, d3, d4}, scale_(scale),
        distortion_(std::fabs(d0) > 0.0000001) {}
  Vector3d cam2world(const double &u, const double &v) const override {
    Vector3d xyz;
    if (!distortion_) { xyz[0] = (u - cx_) / fx_; xyz[1] = (v - cy_) / fy_; xyz[2] = 1.0; }
    else {
      const double uf = (double)(float)u, vf = (double)(float)v;
      const double ifx = 1.0 / fx_, ify = 1.0 / fy_;
      const double x0 = (uf - cx_) * ifx, y0 = (vf - cy_) * ify;
      double x = x0, y = y0;
      for (int j = 0; j < 5; j++) {
        const double r2 = x * x + y * y;
        const double icdist = 1.0 / (1.0 + ((d_[4] * r2 + d_[1]) * r2 + d_[0]) * r2);
        if (icdist < 0) { x = x0; y = y0; break; }
        const double deltaX = 2.0 * d_[2] * x * y + d_[3] * (r2 + 2.0 * x * x);
        const double deltaY = d_[2] * (r2 + 2.0 * y * y) + 2.0 * d_[3] * x * y;
        x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
      }
      xyz[0] = (double)(float)x; xyz[1] = (double)(float)y; xyz[2] = 1.0;
    }
    return xyz.normalized();
  }
  Vector3d cam2world(const Vector2d &px) const override { return cam2world(px[0], px[1]); }
  Vector2d world2cam(const Vector3d &xyz_c) const override { return world2cam(Vector2d(xyz_c[0] / xyz_c[2], xyz_c[1] / xyz_c[2])); }
  Vector2d world2cam(const Vector2d &uv) const override {
    Vector2d px;
    if (!distortion_) { px[0] = fx_ * uv[0] + cx_; px[1] = fy_ * uv[1] + cy_; }
    else {
      double x = uv[0], y = uv[1], r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
      double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
      double cdist = 1 + d_[0] * r2 + d_[1] * r4 + d_[4] * r6;
      double xd = x * cdist + d_[2] * a1 + d_[3] * a2;
      double yd = y * cdist + d_[2] * a3 + d_[3] * a1;
      px[0] = xd * fx_ + cx_; px[1] = yd * fy_ + cy_;
    }
    return px;
  }
  double errorMultiplier2() const override { return std::fabs(fx_); }
  double errorMultiplier() const override { return std::fabs(4.0 * fx_ * fy_); }
  double fx() const override { return fx_; }
  double fy() const override { return fy_; }
  double cx() const override { return cx_; }
  double cy() const override { return cy_; }
  double scale() const override { return scale_; }
  void undistortImage(const cv::Mat &raw, cv::Mat &rectified) { rectified = raw.clone(); }
};
} // namespace vk
