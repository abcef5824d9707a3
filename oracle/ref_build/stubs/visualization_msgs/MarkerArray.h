// TEST INFRASTRUCTURE ONLY — stand-in for visualization_msgs/MarkerArray.h
#pragma once
#include <vector>
#include <visualization_msgs/Marker.h>
namespace visualization_msgs { struct MarkerArray { std::vector<Marker> markers; }; }
