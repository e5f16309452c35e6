// TEST INFRASTRUCTURE ONLY -- stand-in for pcl::PointCloud<T>: a vector of points with the members the reference's
// tools.hpp / voxel_map.hpp call (points, push_back, size, clear, reserve, swap, Ptr, makeShared).  Not PCL.
#pragma once
#include <memory>
#include <vector>
namespace pcl {
template <typename PointT> class PointCloud {
 public:
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  void push_back(const PointT& p) { points.push_back(p); width = (unsigned)points.size(); }
  size_t size() const { return points.size(); }
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = 0; }
  void reserve(size_t n) { points.reserve(n); }
  void resize(size_t n) { points.resize(n); width = (unsigned)n; }
  void swap(PointCloud& o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
  typename std::vector<PointT>::iterator begin() { return points.begin(); }
  typename std::vector<PointT>::iterator end() { return points.end(); }
  PointCloud& operator+=(const PointCloud& o) { points.insert(points.end(), o.points.begin(), o.points.end()); width = (unsigned)points.size(); return *this; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }
};
}  // namespace pcl
