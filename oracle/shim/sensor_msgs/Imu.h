// TEST INFRASTRUCTURE ONLY -- stand-in for sensor_msgs::Imu with the fields IMU_PRE::push_imu reads
// (preintegration.hpp:50-73): header.stamp.toSec(), angular_velocity, linear_acceleration.  Not ROS.
#pragma once
#include <memory>
#include "../ros/ros.h"
namespace sensor_msgs {
struct Vec3 { double x = 0, y = 0, z = 0; };
struct Header { ros::Time stamp; };
struct Imu {
  Header header;
  Vec3 angular_velocity, linear_acceleration;
};
typedef std::shared_ptr<Imu> ImuPtr;
typedef std::shared_ptr<const Imu> ImuConstPtr;
}  // namespace sensor_msgs
