// TEST INFRASTRUCTURE ONLY — stand-in for <vikit/vision.h> (see abstract_camera.h).  THIRD-PARTY, UNPINNED.
// interpolateMat_8u: the library's published float bilinear sample.  shiTomasiScore / halfSample are named by code outside the parity
// path (vio.cpp:822,845; frame.cpp:61) and are declared only.
#pragma once
#include <cmath>
#include <opencv2/opencv.hpp>
namespace vk {
inline float interpolateMat_8u(const cv::Mat &mat, float u, float v) {
  int x = (int)std::floor(u), y = (int)std::floor(v);
  float subpix_x = u - x, subpix_y = v - y;
  float w00 = (1.0f - subpix_x) * (1.0f - subpix_y);
  float w01 = (1.0f - subpix_x) * subpix_y;
  float w10 = subpix_x * (1.0f - subpix_y);
  float w11 = 1.0f - w00 - w01 - w10;
  const int stride = (int)mat.step.p[0];
  unsigned char *ptr = mat.data + y * stride + x;
  return w00 * ptr[0] + w01 * ptr[stride] + w10 * ptr[1] + w11 * ptr[stride + 1];
}
float shiTomasiScore(const cv::Mat &img, int u, int v);
void halfSample(const cv::Mat &in, cv::Mat &out);
} // namespace vk
