// TEST INFRASTRUCTURE ONLY -- stand-in for the one ROS call on the hot path: ros::Time::now().toSec() (timing printouts in
// LI_BA_Optimizer::damping_iter, voxel_map.hpp:585-587).  Not ROS.
#pragma once
#include <chrono>
namespace ros {
class Time {
  double s_ = 0;
 public:
  Time() {}
  explicit Time(double s) : s_(s) {}
  static Time now() { return Time(std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count()); }
  double toSec() const { return s_; }
  Time& fromSec(double s) { s_ = s; return *this; }
};
}  // namespace ros
