// TEST INFRASTRUCTURE ONLY — stand-in for <ros/ros.h> (ROS1 is absent from this image): just enough surface for the declarations and the
// publishing / parameter-loading functions of voxel_map.cpp to compile.  Nothing here is executed by the parity tests.
#pragma once
#include <string>
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
#include <vector>
namespace ros {
struct Time { Time() {} static Time now() { return Time(); } double toSec() const { return 0.0; } Time &fromSec(double) { return *this; } };
struct Duration { Duration() {} explicit Duration(double) {} };
struct Rate { explicit Rate(double) {} void sleep() {} };
class Publisher { public: template <class M> void publish(const M &) const {} int getNumSubscribers() const { return 0; } };
class NodeHandle {
public:
  template <class T> bool param(const std::string &, T &var, const T &def) const { var = def; return false; }
};
} // namespace ros
#define ROS_ERROR(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_ASSERT(c) ((void)0)
