// TEST INFRASTRUCTURE ONLY — stand-in for <pcl/filters/voxel_grid.h>: declared because vio.h includes it; the compiled sources never call it.
#pragma once
#include <pcl/point_cloud.h>
namespace pcl { template <class PointT> class VoxelGrid { public: void setLeafSize(float, float, float) {} }; }
