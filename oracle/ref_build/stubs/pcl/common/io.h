// TEST INFRASTRUCTURE ONLY — stand-in for <pcl/common/io.h>
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
