// TEST INFRASTRUCTURE ONLY — stand-in for sensor_msgs/Imu.h (type named by common_lib.h:66)
#pragma once
#include <memory>
#include <string>
#include <geometry_msgs/Quaternion.h>
namespace std_msgs { struct Header { unsigned seq = 0; struct { double toSec() const { return t; } double t = 0; } stamp; std::string frame_id; }; }
namespace sensor_msgs { struct Imu;
typedef std::shared_ptr<const Imu> ImuConstPtr; typedef std::shared_ptr<Imu> ImuPtr;
struct Imu { typedef std::shared_ptr<Imu> Ptr; typedef std::shared_ptr<const Imu> ConstPtr; std_msgs::Header header; geometry_msgs::Vector3 angular_velocity, linear_acceleration; geometry_msgs::Quaternion orientation; }; }
