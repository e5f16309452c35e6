// vxba_lidar_factor.hpp -- header-only C++ adapter: the reference's `class LidarFactor` interface
// (VoxelSLAM/src/voxel_map.hpp:109-290) and `Lidar_BA_Optimizer::damping_iter` (:367-442) on top of the C ABI
// (vxba.h / libvxba.so), so the reference's call sites compile unchanged against the MI355X path:
//
//   producers : OctoTree::tras_opt  -> push_voxel(...)                 voxel_map.hpp:1321, loop_refine.hpp:387
//   sweeps    : acc_evaluate2 / evaluate_only_residual                  voxel_map.hpp:321,327,352,359,488,516,542,555
//   consumers : pcr_adds[k] / eig_values[k] / eig_vectors[k] / plvec_voxels.size() / win_size
//               voxel_map.hpp:1211-1222, voxelslam.cpp:623,630,651,1609
//
// It is a template over the caller's own types (PointCluster, IMUST, Eigen::Vector3d/Matrix3d/MatrixXd/VectorXd), and
// only needs what Eigen dense objects offer: `operator()(r,c)` / `operator[](i)` / `.data()` (column-major) /
// `.resize()`.  INTEGRATION.md shows the three-line change to voxel_map.hpp.  No CPU fallback: every failure of the
// C ABI throws std::runtime_error with vxba_last_error().
#pragma once
#include <cstddef>
#include <stdexcept>
#include <string>
#include <vector>

#include "vxba.h"

namespace vxba {

template <class PointClusterT, class StateT, class Vec3T, class Mat3T, class MatXT, class VecXT>
class LidarFactorT {
 public:
  // `plvec_voxels.size()` is the only use callers make of plvec_voxels (voxel_map.hpp:314,344,481,533)
  struct SizeProxy {
    const LidarFactorT* f;
    size_t size() const { return (size_t)vxba_size(f->h_); }
  };
  // lazily mirrored read-only views of the cache members (valid until the next sweep / push / clear)
  template <class T>
  struct CacheView {
    LidarFactorT* f;
    std::vector<T>* store;
    const T& operator[](size_t k) const { f->sync_cache(); return (*store)[k]; }
    size_t size() const { return (size_t)vxba_size(f->h_); }
  };

  SizeProxy plvec_voxels{this};
  CacheView<PointClusterT> pcr_adds{this, &pcr_adds_};
  CacheView<Vec3T> eig_values{this, &eig_values_};
  CacheView<Mat3T> eig_vectors{this, &eig_vectors_};
  int win_size;

  explicit LidarFactorT(int w, int device = 0) : win_size(w) {
    if (vxba_create(w, device, &h_) != VXBA_OK) throw std::runtime_error("vxba_create failed (needs a gfx950 GPU, win_size <= 128)");
  }
  ~LidarFactorT() { vxba_destroy(h_); }
  LidarFactorT(const LidarFactorT&) = delete;
  LidarFactorT& operator=(const LidarFactorT&) = delete;

  // voxel_map.hpp:122-130.  Voxels are staged on the host and uploaded in one batch by the next sweep.
  void push_voxel(std::vector<PointClusterT>& vec_orig, PointClusterT& fix, double coe, Vec3T& eig_value, Mat3T& eig_vector,
                  PointClusterT& pcr_add) {
    apply_win_size();
    for (int i = 0; i < win_size; i++) pack_cluster(vec_orig[i], st_clusters_);
    pack_cluster(fix, st_fix_);
    st_coe_.push_back(coe);
    for (int k = 0; k < 3; k++) st_eigval_.push_back(eig_value[k]);
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) st_eigvec_.push_back(eig_vector(r, c));
    pack_cluster(pcr_add, st_merged_);
    cache_valid_ = false;
  }

  // voxel_map.hpp:132-241
  void acc_evaluate2(const std::vector<StateT>& xs, int head, int end, MatXT& Hess, VecXT& JacT, double& residual) {
    flush();
    pack_poses(xs);
    check(vxba_acc_evaluate2(h_, rp_.data(), head, end, Hess.data(), JacT.data(), &residual));
  }

  // voxel_map.hpp:243-279
  void evaluate_only_residual(const std::vector<StateT>& xs, int head, int end, double& residual) {
    flush();
    pack_poses(xs);
    check(vxba_evaluate_only_residual(h_, rp_.data(), head, end, &residual));
    cache_valid_ = false;
  }

  // voxel_map.hpp:281-286
  void clear() {
    check(vxba_clear(h_));
    st_clusters_.clear(); st_fix_.clear(); st_coe_.clear(); st_eigval_.clear(); st_eigvec_.clear(); st_merged_.clear();
    cache_valid_ = false;
  }

  // Lidar_BA_Optimizer::damping_iter (voxel_map.hpp:367-442) with the whole loop resident on the GPU.
  bool damping_iter(std::vector<StateT>& x_stats, MatXT* hess, std::vector<double>& resis, int max_iter = 3) {
    flush();
    pack_poses(x_stats);
    const int n = 6 * win_size;
    hess->resize(n, n);
    double rs[2] = {0, 0};
    int conv = 0, nt = 0;
    check(vxba_damping_iter(h_, rp_.data(), max_iter, hess->data(), rs, nullptr, &nt, &conv));
    for (int i = 0; i < win_size; i++) {
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) x_stats[i].R(r, c) = rp_[12 * i + 3 * c + r];
      for (int k = 0; k < 3; k++) x_stats[i].p[k] = rp_[12 * i + 9 + k];
    }
    resis.push_back(rs[0]);
    resis.push_back(rs[1]);
    cache_valid_ = false;
    return conv != 0;
  }

  vxba_factor* handle() { flush(); return h_; }

 private:
  friend struct SizeProxy;
  vxba_factor* h_ = nullptr;
  std::vector<double> st_clusters_, st_fix_, st_coe_, st_eigval_, st_eigvec_, st_merged_, rp_;
  std::vector<PointClusterT> pcr_adds_;
  std::vector<Vec3T> eig_values_;
  std::vector<Mat3T> eig_vectors_;
  bool cache_valid_ = false;

  void check(int rc) {
    if (rc != VXBA_OK) throw std::runtime_error(std::string("vxba: ") + vxba_last_error(h_));
  }
  void apply_win_size() {  // `voxhess.win_size = ...` is a plain member assignment upstream (voxelslam.cpp:623,1609)
    if (win_size != vxba_win_size(h_)) check(vxba_set_win_size(h_, win_size));
  }
  static void pack_cluster(const PointClusterT& pc, std::vector<double>& out) {
    out.push_back(pc.P(0, 0)); out.push_back(pc.P(0, 1)); out.push_back(pc.P(0, 2));
    out.push_back(pc.P(1, 1)); out.push_back(pc.P(1, 2)); out.push_back(pc.P(2, 2));
    out.push_back(pc.v[0]); out.push_back(pc.v[1]); out.push_back(pc.v[2]);
    out.push_back((double)pc.N);
  }
  void pack_poses(const std::vector<StateT>& xs) {  // R, p extracted field-wise -- never memcpy the IMUST struct
    rp_.resize((size_t)12 * win_size);
    for (int i = 0; i < win_size; i++) {
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) rp_[12 * i + 3 * c + r] = xs[i].R(r, c);
      for (int k = 0; k < 3; k++) rp_[12 * i + 9 + k] = xs[i].p[k];
    }
  }
  void flush() {
    apply_win_size();
    const int n = (int)st_coe_.size();
    if (n == 0) return;
    check(vxba_push_voxels(h_, n, st_clusters_.data(), st_fix_.data(), st_coe_.data(), st_eigval_.data(), st_eigvec_.data(),
                           st_merged_.data()));
    st_clusters_.clear(); st_fix_.clear(); st_coe_.clear(); st_eigval_.clear(); st_eigvec_.clear(); st_merged_.clear();
  }
  void sync_cache() {
    if (cache_valid_) return;
    flush();
    const int n = vxba_size(h_);
    std::vector<double> ev((size_t)3 * n), U((size_t)9 * n), m((size_t)10 * n);
    check(vxba_read_cache(h_, 0, n, ev.data(), U.data(), m.data()));
    pcr_adds_.resize(n); eig_values_.resize(n); eig_vectors_.resize(n);
    for (int a = 0; a < n; a++) {
      for (int k = 0; k < 3; k++) eig_values_[a][k] = ev[3 * a + k];
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) eig_vectors_[a](r, c) = U[9 * a + 3 * c + r];
      const double* c = &m[10 * (size_t)a];
      PointClusterT& pc = pcr_adds_[a];
      pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
      pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
      pc.v[0] = c[6]; pc.v[1] = c[7]; pc.v[2] = c[8];
      pc.N = (int)c[9];
    }
    cache_valid_ = true;
  }
};

}  // namespace vxba
