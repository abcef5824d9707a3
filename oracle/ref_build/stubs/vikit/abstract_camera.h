// TEST INFRASTRUCTURE ONLY — stand-in for rpg_vikit (xuankuzcr fork, no version pinned: reference README.md:76-84), absent from this image.
// Interface = the members the reference calls on vk::AbstractCamera (vio.cpp:45-54, 109, 261-282, 1447, 1574; frame.h:49-66).
// THIRD-PARTY, UNPINNED: the arithmetic in pinhole_camera.h is the library's published pinhole / radial-tangential model, not reference code.
#pragma once
#include <Eigen/Dense>
#include <opencv2/opencv.hpp>
namespace vk {
using Eigen::Vector2d; using Eigen::Vector2i; using Eigen::Vector3d;
class AbstractCamera {
protected:
  int width_ = 0, height_ = 0;
public:
  AbstractCamera() {}
  AbstractCamera(int w, int h) : width_(w), height_(h) {}
  virtual ~AbstractCamera() {}
  virtual Vector3d cam2world(const double &x, const double &y) const = 0;
  virtual Vector3d cam2world(const Vector2d &px) const = 0;
  virtual Vector2d world2cam(const Vector3d &xyz_c) const = 0;
  virtual Vector2d world2cam(const Vector2d &uv) const = 0;
  virtual double errorMultiplier2() const = 0;
  virtual double errorMultiplier() const = 0;
  virtual double fx() const = 0;
  virtual double fy() const = 0;
  virtual double cx() const = 0;
  virtual double cy() const = 0;
  virtual double scale() const { return 1.0; }
  int width() const { return width_; }
  int height() const { return height_; }
  bool isInFrame(const Vector2i &obs, int boundary = 0) const {
    return obs[0] >= boundary && obs[0] < width() - boundary && obs[1] >= boundary && obs[1] < height() - boundary;
  }
};
} // namespace vk
