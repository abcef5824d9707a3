// TEST INFRASTRUCTURE ONLY — stand-in for geometry_msgs/Quaternion.h
#pragma once
namespace geometry_msgs { struct Quaternion { double x = 0, y = 0, z = 0, w = 1; }; struct Point { double x = 0, y = 0, z = 0; }; struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; }; }
