// TEST INFRASTRUCTURE ONLY -- stand-in for pcl::KdTreeFLANN as loop_refine.hpp's icp_normal declares it (out of scope, never called
// through oracle/ref_capi.cpp): a brute-force nearest neighbour so that the header compiles and links.  Not PCL.
#pragma once
#include <vector>
#include "../point_cloud.h"
#include "../point_types.h"
namespace pcl {
template <typename PointT> class KdTreeFLANN {
  typename PointCloud<PointT>::Ptr cloud_;
 public:
  void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { cloud_ = c; }
  int nearestKSearch(const PointT& p, int k, std::vector<int>& idx, std::vector<float>& d2) const {
    if (!cloud_ || cloud_->empty() || k < 1) return 0;
    int best = 0; float bd = 3.4e38f;
    for (size_t i = 0; i < cloud_->size(); i++) {
      const PointT& q = (*cloud_)[i];
      const float d = (q.x - p.x) * (q.x - p.x) + (q.y - p.y) * (q.y - p.y) + (q.z - p.z) * (q.z - p.z);
      if (d < bd) { bd = d; best = (int)i; }
    }
    idx.assign(1, best); d2.assign(1, bd);
    return 1;
  }
};
}  // namespace pcl
