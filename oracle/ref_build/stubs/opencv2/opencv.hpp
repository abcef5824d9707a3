// TEST INFRASTRUCTURE ONLY — stand-in for <opencv2/opencv.hpp> (OpenCV is absent from this image).  cv::Mat is a reference-counted
// byte buffer with rows/cols/type, exactly what the compiled reference sources read (`img.data`, `img.cols`, `clone()`, `zeros`); the
// drawing / file / colour-conversion functions named by the plotting code exist as no-ops so that vio.cpp compiles whole — the parity
// tests never reach them.
#pragma once
#include <cstdint>
#include <algorithm>
#include <deque>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <list>
#include <map>
#include <numeric>
#include <set>
#include <sstream>
#include <unordered_map>
#include <cstring>
#include <memory>
#include <string>
#include <vector>
typedef unsigned char uchar;
#define CV_8U 0
#define CV_32F 5
#define CV_MAKETYPE(depth, cn) ((depth) + (((cn)-1) << 3))
#define CV_8UC1 CV_MAKETYPE(CV_8U, 1)
#define CV_8UC3 CV_MAKETYPE(CV_8U, 3)
#define CV_32FC1 CV_MAKETYPE(CV_32F, 1)
#define CV_32FC2 CV_MAKETYPE(CV_32F, 2)
namespace cv {
struct Size { int width = 0, height = 0; Size() {} template <class A, class B> Size(A w, B h) : width((int)w), height((int)h) {} };
struct Point2f { float x = 0, y = 0; Point2f() {} template <class A, class B> Point2f(A a, B b) : x((float)a), y((float)b) {} };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {} };
enum { FONT_HERSHEY_COMPLEX = 3 };
class Mat {
  std::shared_ptr<std::vector<uchar>> buf_;
  int type_ = 0;
  static int elem(int type) { const int depth = type & 7, cn = (type >> 3) + 1; return (depth == CV_32F ? 4 : 1) * cn; }
public:
  uchar *data = nullptr;
  int rows = 0, cols = 0;
  struct Step { size_t p[2] = {0, 0}; operator size_t() const { return p[0]; } } step;
  Mat() {}
  Mat(int r, int c, int type) { create(r, c, type); }
  Mat(int r, int c, int type, void *ext) : type_(type), data((uchar *)ext), rows(r), cols(c) { step.p[0] = (size_t)c * elem(type); step.p[1] = elem(type); }   // external data, not owned
  void create(int r, int c, int type) { buf_ = std::make_shared<std::vector<uchar>>((size_t)r * c * elem(type), (uchar)0); data = buf_->data(); rows = r; cols = c; type_ = type; step.p[0] = (size_t)c * elem(type); step.p[1] = elem(type); }
  static Mat zeros(int r, int c, int type) { return Mat(r, c, type); }
  Mat clone() const { Mat m; if (!data) return m; m.create(rows, cols, type_); std::memcpy(m.data, data, (size_t)rows * cols * elem(type_)); return m; }
  bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
  int type() const { return type_; }
  int channels() const { return (type_ >> 3) + 1; }
  template <class T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step.p[0] + (size_t)c * sizeof(T)); }
  template <class T> T *ptr(int r = 0) { return (T *)(data + (size_t)r * step.p[0]); }
};
inline void circle(Mat &, Point2f, int, Scalar, int = 1, int = 8, int = 0) {}
inline void line(Mat &, Point2f, Point2f, Scalar, int = 1, int = 8, int = 0) {}
inline void rectangle(Mat &, Point2f, Point2f, Scalar, int = 1, int = 8, int = 0) {}
inline void putText(Mat &, const std::string &, Point2f, int, double, Scalar, int = 1, int = 8, bool = false) {}
inline void hconcat(const Mat &, const Mat &, Mat &) {}
inline void cvtColor(const Mat &, Mat &, int) {}
inline void absdiff(const Mat &, const Mat &, Mat &) {}
inline void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1) {}
inline bool imwrite(const std::string &, const Mat &) { return false; }
inline void imshow(const std::string &, const Mat &) {}
inline int waitKey(int = 0) { return -1; }
} // namespace cv
