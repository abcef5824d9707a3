// TEST INFRASTRUCTURE ONLY — stand-in for <pcl/point_types.h> (PCL is absent from this image): the point structs the reference names, as
// plain aggregates with PCL's documented field names.  DEG2RAD is PCL's macro from pcl/pcl_macros.h (used by voxel_map.cpp:21):
// its published definition is ((x)*0.017453293) — a 9-digit constant, not M_PI/180.  oracle/orc_lidar.hpp takes the same value as `deg2rad`.
#pragma once
#include <cstdint>
#ifndef DEG2RAD
#define DEG2RAD(x) ((x)*0.017453293)
#endif
#ifndef RAD2DEG
#define RAD2DEG(x) ((x)*57.29578)
#endif
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0; };
struct PointXYZI { float x = 0, y = 0, z = 0; float intensity = 0; };
struct PointXYZINormal { float x = 0, y = 0, z = 0; float normal_x = 0, normal_y = 0, normal_z = 0; float intensity = 0; float curvature = 0; };
struct PointXYZRGB { float x = 0, y = 0, z = 0; std::uint8_t b = 0, g = 0, r = 0, a = 255; };
struct PointXYZRGBA { float x = 0, y = 0, z = 0; std::uint8_t b = 0, g = 0, r = 0, a = 255; };
} // namespace pcl
