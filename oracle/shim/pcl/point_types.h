// TEST INFRASTRUCTURE ONLY -- stand-in for the PCL point type the reference's tools.hpp names (pcl::PointXYZINormal):
// the fields the three hot-path headers touch, nothing of PCL itself.  See oracle/shim/Eigen/Core for why this exists.
#pragma once
namespace pcl {
struct PointXYZ {
  union { float data[4]; struct { float x, y, z; }; };
  PointXYZ() : data{0, 0, 0, 1.f} {}
};
struct PointXYZINormal {
  union { float data[4]; struct { float x, y, z; }; };
  union { float data_n[4]; float normal[3]; struct { float normal_x, normal_y, normal_z; }; };
  union { struct { float intensity; float curvature; }; float data_c[4]; };
  PointXYZINormal() : data{0, 0, 0, 1.f}, data_n{0, 0, 0, 0}, data_c{0, 0, 0, 0} {}
};
}  // namespace pcl
