// TEST INFRASTRUCTURE ONLY -- stand-ins for the GTSAM types loop_refine.hpp names in signatures next to OctreeGBA (the pose-graph
// code itself is out of scope and never called through oracle/ref_capi.cpp).  Not GTSAM.  See oracle/shim/Eigen/Core.
#pragma once
#include <memory>
#include <vector>
#include <Eigen/Core>
namespace gtsam {
typedef unsigned long Key;
class Point3 : public Eigen::Vector3d {
 public:
  Point3() {}
  Point3(const Eigen::Vector3d& v) : Eigen::Vector3d(v) {}
  Point3(double x, double y, double z) : Eigen::Vector3d(x, y, z) {}
};
class Rot3 {
  Eigen::Matrix3d R_;
 public:
  Rot3() { R_.setIdentity(); }
  Rot3(const Eigen::Matrix3d& R) : R_(R) {}
  const Eigen::Matrix3d& matrix() const { return R_; }
};
class Pose3 {
  Rot3 R_;
  Point3 t_;
 public:
  Pose3() {}
  Pose3(const Rot3& R, const Point3& t) : R_(R), t_(t) {}
  const Rot3& rotation() const { return R_; }
  const Point3& translation() const { return t_; }
};
class NonlinearFactor {
 public:
  typedef std::shared_ptr<NonlinearFactor> shared_ptr;
  virtual ~NonlinearFactor() {}
};
namespace noiseModel {
class Diagonal {
 public:
  typedef std::shared_ptr<Diagonal> shared_ptr;
};
}  // namespace noiseModel
class NonlinearFactorGraph {
  std::vector<NonlinearFactor::shared_ptr> f_;
 public:
  void push_back(const NonlinearFactor::shared_ptr& f) { f_.push_back(f); }
  size_t size() const { return f_.size(); }
};
}  // namespace gtsam
