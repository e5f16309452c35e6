// vxba_lio_estimator.hpp -- header-only C++ adapter for the odometry's point-to-plane state estimation on top of the C ABI (vxba.h):
// `lio_state_estimation` (VoxelSLAM/src/voxelslam.cpp:855-958) against the voxel plane map `surf_map`
// (unordered_map<VOXEL_LOC, OctoTree*>, voxel_map.hpp:925-1500), with `pvec_update` (voxelslam.hpp:203-215).
//
//   call sites replaced:   if(lio_state_estimation(pptr)) ...          voxelslam.cpp:1248, 1587
//                          pvec_update(pptr, x_curr, pwld);            voxelslam.cpp:1592
//   call sites added:      after the per-scan map update (cut_voxel / recut / margi, voxelslam.cpp:1398-1453), for every root voxel
//                          the scan touched:   est.stage_voxel(loc, octree);   and once:   est.flush_map();
//
// Template over the caller's own types -- only field names and operator()(r,c) / operator[](i) are used:
//   StateT    = IMUST      R, p, v, bg, ba, g, cov (15x15)                     tools.hpp:135-199
//   PointVarT = pointVar   pnt (3), var (3x3)                                  voxel_map.hpp:14-19
//   OctoTreeT = OctoTree   layer, octo_state, leaves[8], plane.{center, normal, plane_var (6x6), radius, is_plane}
//   VoxelLocT = VOXEL_LOC  x, y, z (int64)                                     tools.hpp:24-35
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "vxba.h"

namespace vxba {

template <class StateT, class PointVarT>
class LioEstimatorT {
 public:
  LioEstimatorT(double voxel_size, int max_layer, int device = 0) {
    if (vxba_lio_create(voxel_size, max_layer, device, &h_) != VXBA_OK) throw std::runtime_error("vxba_lio_create failed (an MI355X is required; no CPU fallback)");
  }
  ~LioEstimatorT() { if (h_) vxba_lio_destroy(h_); }
  LioEstimatorT(const LioEstimatorT&) = delete;
  LioEstimatorT& operator=(const LioEstimatorT&) = delete;

  // Walk one root voxel of surf_map the way OctoTree::match does and stage every leaf (octo_state == 0) for upload.  A leaf whose
  // plane is not valid (is_plane == false) is staged as a removal, so a leaf that lost its plane stops matching.
  template <class VoxelLocT, class OctoTreeT>
  void stage_voxel(const VoxelLocT& loc, const OctoTreeT* node, int path = 0) {
    if (node->octo_state == 0) {
      loc_.push_back((int64_t)loc.x); loc_.push_back((int64_t)loc.y); loc_.push_back((int64_t)loc.z);
      layer_.push_back((int32_t)node->layer); path_.push_back((int32_t)path); is_plane_.push_back(node->plane.is_plane ? 1 : 0);
      for (int k = 0; k < 3; k++) { center_.push_back(node->plane.center[k]); normal_.push_back(node->plane.normal[k]); }
      for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) plane_var_.push_back(node->plane.plane_var(r, c));
      radius_.push_back((double)node->plane.radius);
      return;
    }
    for (int i = 0; i < 8; i++)
      if (node->leaves[i]) stage_voxel(loc, node->leaves[i], path | (i << (3 * node->layer)));
  }
  // A node the host tree has just subdivided: its old leaf entry must go before the children arrive (earlier flush).
  template <class VoxelLocT>
  void stage_removal(const VoxelLocT& loc, int layer, int path) {
    loc_.push_back((int64_t)loc.x); loc_.push_back((int64_t)loc.y); loc_.push_back((int64_t)loc.z);
    layer_.push_back(layer); path_.push_back(path); is_plane_.push_back(0);
    for (int k = 0; k < 3; k++) { center_.push_back(0.0); normal_.push_back(0.0); }
    for (int k = 0; k < 36; k++) plane_var_.push_back(0.0);
    radius_.push_back(0.0);
  }
  void flush_map() {
    const int64_t n = (int64_t)layer_.size();
    if (n) check(vxba_lio_map_update(h_, n, loc_.data(), layer_.data(), path_.data(), is_plane_.data(), center_.data(), normal_.data(), plane_var_.data(), radius_.data()));
    loc_.clear(); layer_.clear(); path_.clear(); is_plane_.clear(); center_.clear(); normal_.clear(); plane_var_.clear(); radius_.clear();
  }
  void clear_map() { check(vxba_lio_map_clear(h_)); }

  // pptr: the scan as var_init left it (IMU-frame points with their covariances)
  void set_scan(const std::vector<PointVarT>& pvec) {
    const size_t n = pvec.size();
    pnt_.resize(3 * n); var_.resize(9 * n);
    for (size_t i = 0; i < n; i++) {
      for (int k = 0; k < 3; k++) pnt_[3 * i + k] = pvec[i].pnt[k];
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) var_[9 * i + 3 * c + r] = pvec[i].var(r, c);
    }
    check(vxba_lio_scan_set(h_, (int64_t)n, pnt_.data(), var_.data()));
  }
  // var_init itself on the GPU: sensor-frame float points + extrinsic
  template <class ExtT>
  void set_scan_raw(const float* xyz, int64_t n, const ExtT& ext, double dept_err, double beam_err) {
    double e[12];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) e[3 * c + r] = ext.R(r, c);
    for (int k = 0; k < 3; k++) e[9 + k] = ext.p[k];
    check(vxba_lio_scan_raw(h_, n, xyz, e, dept_err, beam_err));
  }

  // bool lio_state_estimation(PVecPtr pptr) with x_curr passed explicitly (in/out, including cov)
  bool lio_state_estimation(StateT& x_curr) {
    double st[VXBA_STATE_LEN], cov[225], info[4];
    pack(x_curr, st, cov);
    check(vxba_lio_state_estimation(h_, st, cov, info, nullptr));
    unpack(st, cov, x_curr);
    match_num = (int)info[2]; iterations = (int)info[1]; min_eigen = info[3];
    return info[0] != 0.0;
  }
  // pvec_update: world points and world covariances of the scan under x_curr
  template <class V3>
  void pvec_update(const StateT& x_curr, std::vector<PointVarT>& world_var, std::vector<V3>& pwld) {
    double st[VXBA_STATE_LEN], cov[225];
    pack(x_curr, st, cov);
    const size_t n = (size_t)vxba_lio_scan_size(h_);
    pnt_.resize(3 * n); var_.resize(9 * n);
    check(vxba_lio_pvec_update(h_, st, cov, pnt_.data(), var_.data()));
    world_var.resize(n); pwld.resize(n);
    for (size_t i = 0; i < n; i++) {
      for (int k = 0; k < 3; k++) { pwld[i][k] = pnt_[3 * i + k]; world_var[i].pnt[k] = pnt_[3 * i + k]; }
      for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) world_var[i].var(r, c) = var_[9 * i + 3 * c + r];
    }
  }
  // the same, world points only: the covariances stay on the device for leaf_stats (72 of the 96 bytes per point never cross PCIe)
  template <class V3>
  void pvec_update_points(const StateT& x_curr, std::vector<V3>& pwld) {
    double st[VXBA_STATE_LEN], cov[225];
    pack(x_curr, st, cov);
    const size_t n = (size_t)vxba_lio_scan_size(h_);
    pnt_.resize(3 * n);
    check(vxba_lio_pvec_update(h_, st, cov, pnt_.data(), nullptr));
    pwld.resize(n);
    for (size_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) pwld[i][k] = pnt_[3 * i + k];
  }
  // What cut_voxel adds to the leaves the scan touched (OctoTree::push, voxel_map.hpp:969-993): leaf c owns the scan points
  // order[cell_ptr[c] .. cell_ptr[c+1]) in push order.  clusters[10c..] = P (xx xy xz yy yz zz), v, N of those points -- add to
  // pcr_add; cov_add[81c..] (col-major 9x9) -- add to the node's cov_add.
  void leaf_stats(const std::vector<int64_t>& cell_ptr, const std::vector<int32_t>& order, std::vector<double>& clusters, std::vector<double>& cov_add) {
    if (cell_ptr.empty()) throw std::invalid_argument("leaf_stats: cell_ptr needs n_cells + 1 entries");
    const size_t n = cell_ptr.size() - 1;
    if ((int64_t)order.size() != cell_ptr.back()) throw std::invalid_argument("leaf_stats: order must hold cell_ptr.back() indices");
    clusters.assign(10 * n, 0.0); cov_add.assign(81 * n, 0.0);
    check(vxba_lio_leaf_stats(h_, (int64_t)n, cell_ptr.data(), order.data(), clusters.data(), cov_add.data()));
  }

  int match_num = 0, iterations = 0;
  double min_eigen = 0.0;
  vxba_lio* handle() { return h_; }

 private:
  static void pack(const StateT& x, double* st, double* cov) {
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) st[3 * c + r] = x.R(r, c);
    for (int k = 0; k < 3; k++) { st[9 + k] = x.p[k]; st[12 + k] = x.v[k]; st[15 + k] = x.bg[k]; st[18 + k] = x.ba[k]; st[21 + k] = x.g[k]; }
    for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) cov[15 * c + r] = x.cov(r, c);
  }
  static void unpack(const double* st, const double* cov, StateT& x) {
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) x.R(r, c) = st[3 * c + r];
    for (int k = 0; k < 3; k++) { x.p[k] = st[9 + k]; x.v[k] = st[12 + k]; x.bg[k] = st[15 + k]; x.ba[k] = st[18 + k]; }
    for (int c = 0; c < 15; c++) for (int r = 0; r < 15; r++) x.cov(r, c) = cov[15 * c + r];
  }
  void check(int rc) { if (rc != VXBA_OK) throw std::runtime_error(std::string("vxba_lio: ") + vxba_lio_last_error(h_)); }

  vxba_lio* h_ = nullptr;
  std::vector<int64_t> loc_;
  std::vector<int32_t> layer_, path_, is_plane_;
  std::vector<double> center_, normal_, plane_var_, radius_, pnt_, var_;
};

}  // namespace vxba
