// TEST INFRASTRUCTURE ONLY — stand-in for visualization_msgs/Marker.h (fields written by VoxelMapManager::pubSinglePlane, never executed here)
#pragma once
#include <string>
#include <vector>
#include <geometry_msgs/Quaternion.h>
#include <ros/ros.h>
namespace visualization_msgs {
struct Marker {
  enum { CYLINDER = 3, ADD = 0 };
  struct { std::string frame_id; ros::Time stamp; } header;
  std::string ns; int id = 0; int type = 0; int action = 0;
  geometry_msgs::Pose pose; geometry_msgs::Vector3 scale;
  struct { float r = 0, g = 0, b = 0, a = 0; } color;
  ros::Duration lifetime;
};
} // namespace visualization_msgs
