// TEST INFRASTRUCTURE ONLY — stand-in for <pcl/point_cloud.h>: a vector of points with the members the reference uses.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
namespace pcl {
template <class PointT> class PointCloud {
public:
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  std::uint32_t width = 0, height = 0;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void reserve(std::size_t n) { points.reserve(n); }
  void resize(std::size_t n) { points.resize(n); }
  void clear() { points.clear(); }
  void push_back(const PointT &p) { points.push_back(p); }
  void swap(PointCloud &o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); }
  PointT &operator[](std::size_t i) { return points[i]; }
  const PointT &operator[](std::size_t i) const { return points[i]; }
  PointT &at(std::size_t i) { return points.at(i); }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  PointCloud &operator+=(const PointCloud &o) { points.insert(points.end(), o.points.begin(), o.points.end()); return *this; }
};
} // namespace pcl
