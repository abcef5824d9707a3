// TEST INFRASTRUCTURE ONLY — stand-in for nav_msgs/Odometry.h (included by IMU_Processing.h, unused by the compiled sources)
#pragma once
namespace nav_msgs { struct Odometry {}; }
