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
#include <cstdio>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "vxba.h"

namespace vxba {

template <class PointClusterT, class StateT, class Vec3T, class Mat3T, class MatXT, class VecXT>
class LidarFactorT {
 public:
  // The reference's six public vectors, as far as callers touch them:
  //   .size()                                  voxel_map.hpp:314,344,481,533,689,746,1211; voxelslam.cpp:630
  //   [k] on pcr_adds / eig_values / eig_vectors     OctoTree::margi voxel_map.hpp:1217-1221 (through a const LidarFactor&, from several
  //                                                  threads at once: the lazy read-back below is guarded by a mutex)
  //   range-for over eig_vectors               motion_init, voxelslam.cpp:651
  //   a.X.insert(a.X.end(), b.X.begin(), b.X.end()) for all six X     OctreeGBA_multi_recut, loop_refine.hpp:529-534 (b holds staged voxels only)
  //   plvec_voxels[a][i]                       read access to one voxel's window clusters
  enum Member { M_PLVEC = 0, M_SIG, M_COE, M_EIGVAL, M_EIGVEC, M_PCRADD };
  struct MemberIter {          // position inside one member of one factor; only begin()/end() of a whole member are meaningful
    const LidarFactorT* owner; int member; bool at_end;
  };
  struct MemberProxy {
    LidarFactorT* f; int member;
    size_t size() const { return f->member_size(member); }
    MemberIter begin() const { return MemberIter{f, member, false}; }
    MemberIter end() const { return MemberIter{f, member, true}; }
    void insert(MemberIter pos, MemberIter first, MemberIter last) {
      if (pos.owner != f || !pos.at_end || first.at_end || !last.at_end || first.owner != last.owner || first.member != member)
        throw std::runtime_error("vxba: only X.insert(X.end(), other.X.begin(), other.X.end()) is supported");
      f->append_member(member, *first.owner);
    }
  };
  struct PlvecProxy : MemberProxy {
    std::vector<PointClusterT> operator[](size_t a) const { return this->f->voxel_clusters(a); }
  };
  // lazily mirrored views of the cache members (valid until the next sweep / push / clear).  Their iterators know the factor they
  // belong to (for the insert idiom) and only touch the device when dereferenced (so begin()/end() of a factor that merely holds
  // pushed voxels does not upload them).
  template <class T>
  struct CacheIter {
    const LidarFactorT* owner; int member; std::vector<T>* store; size_t idx; bool is_end;
    T& operator*() const { owner->sync_cache(); return (*store)[idx]; }
    T* operator->() const { return &**this; }
    CacheIter& operator++() { ++idx; return *this; }
    bool operator!=(const CacheIter& o) const { return idx != o.idx; }
    bool operator==(const CacheIter& o) const { return idx == o.idx; }
  };
  template <class T>
  struct CacheView {
    LidarFactorT* f; int member; std::vector<T>* store;
    size_t size() const { return f->member_size(member); }
    const T& operator[](size_t k) const { f->sync_cache(); return (*store)[k]; }
    CacheIter<T> begin() const { return CacheIter<T>{f, member, store, 0, false}; }
    CacheIter<T> end() const { return CacheIter<T>{f, member, store, size(), true}; }
    void insert(CacheIter<T> pos, CacheIter<T> first, CacheIter<T> last) {
      if (pos.owner != f || !pos.is_end || first.is_end || !last.is_end || first.owner != last.owner || first.member != member)
        throw std::runtime_error("vxba: only X.insert(X.end(), other.X.begin(), other.X.end()) is supported");
      f->append_member(member, *first.owner);
    }
  };

  PlvecProxy plvec_voxels;
  MemberProxy sig_vecs, coeffs;
  CacheView<PointClusterT> pcr_adds;
  CacheView<Vec3T> eig_values;
  CacheView<Mat3T> eig_vectors;
  int win_size;

  explicit LidarFactorT(int w, int device = 0) : win_size(w), device_(device) {
    bind_proxies();
    if (vxba_create(w, device, &h_) != VXBA_OK) throw std::runtime_error("vxba_create failed (needs a gfx950 GPU, win_size <= 128)");
  }
  ~LidarFactorT() { vxba_destroy(h_); }
  // `vector<LidarFactor> vec_voxhess(thd_num, voxhess)` (loop_refine.hpp:499): a copy is a new device factor holding the same voxels
  LidarFactorT(const LidarFactorT& o) : win_size(o.win_size), device_(o.device_) {
    bind_proxies();
    if (vxba_create(win_size, device_, &h_) != VXBA_OK) throw std::runtime_error("vxba_create failed");
    copy_content(o);
  }
  LidarFactorT& operator=(const LidarFactorT& o) {
    if (this == &o) return *this;
    clear();
    win_size = o.win_size;
    copy_content(o);
    return *this;
  }
  // All six members of `o` appended at once (what the six insert() calls of loop_refine.hpp:529-534 add up to).
  void append(const LidarFactorT& o) { for (int m = 0; m < 6; m++) append_member(m, o); }

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
  bool damping_iter(std::vector<StateT>& x_stats, MatXT* hess, std::vector<double>& resis, int max_iter = 3, bool is_display = false) {
    flush();
    pack_poses(x_stats);
    const int n = 6 * win_size;
    hess->resize(n, n);
    double rs[2] = {0, 0};
    int conv = 0, nt = 0;
    std::vector<double> tr(is_display ? (size_t)VXBA_TRACE_COLS * (max_iter > 0 ? max_iter : 1) : 0);
    check(vxba_damping_iter(h_, rp_.data(), max_iter, hess->data(), rs, is_display ? tr.data() : nullptr, &nt, &conv));
    for (int i = 0; is_display && i < nt; i++) {   // the reference's own line (voxel_map.hpp:415-416)
      const double* t = &tr[(size_t)VXBA_TRACE_COLS * i];
      printf("iter%d: (%lf %lf) u: %lf v: %.1lf q: %.2lf %lf %lf\n", i, t[0], t[1], t[2], t[3], t[4] / t[5], t[5], t[4]);
    }
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
  void invalidate_cache() { cache_valid_ = false; }      // after a sweep issued through handle()

 private:
  vxba_factor* h_ = nullptr;
  int device_ = 0;
  mutable std::mutex mu_;
  std::vector<double> st_clusters_, st_fix_, st_coe_, st_eigval_, st_eigvec_, st_merged_, rp_;
  std::vector<PointClusterT> pcr_adds_;
  std::vector<Vec3T> eig_values_;
  std::vector<Mat3T> eig_vectors_;
  bool cache_valid_ = false;

  void check(int rc) const {
    if (rc != VXBA_OK) throw std::runtime_error(std::string("vxba: ") + vxba_last_error(h_));
  }
  void bind_proxies() {
    plvec_voxels.f = this; plvec_voxels.member = M_PLVEC;
    sig_vecs.f = this; sig_vecs.member = M_SIG;
    coeffs.f = this; coeffs.member = M_COE;
    pcr_adds.f = this; pcr_adds.member = M_PCRADD; pcr_adds.store = &pcr_adds_;
    eig_values.f = this; eig_values.member = M_EIGVAL; eig_values.store = &eig_values_;
    eig_vectors.f = this; eig_vectors.member = M_EIGVEC; eig_vectors.store = &eig_vectors_;
  }
  std::vector<double>& staged(int m) {
    switch (m) { case M_PLVEC: return st_clusters_; case M_SIG: return st_fix_; case M_COE: return st_coe_; case M_EIGVAL: return st_eigval_;
                 case M_EIGVEC: return st_eigvec_; default: return st_merged_; }
  }
  const std::vector<double>& staged(int m) const { return const_cast<LidarFactorT*>(this)->staged(m); }
  size_t per_voxel(int m) const { return m == M_PLVEC ? (size_t)10 * win_size : m == M_SIG ? 10 : m == M_COE ? 1 : m == M_EIGVAL ? 3 : m == M_EIGVEC ? 9 : 10; }
  size_t member_size(int m) const { return (size_t)vxba_size(h_) + staged(m).size() / per_voxel(m); }
  void append_member(int m, const LidarFactorT& o) {
    if (o.win_size != win_size) throw std::runtime_error("vxba: appending a factor with another win_size");
    if (vxba_size(o.h_) != 0) throw std::runtime_error("vxba: the appended factor already uploaded its voxels (append before its first sweep)");
    const std::vector<double>& src = o.staged(m);
    std::vector<double>& dst = staged(m);
    dst.insert(dst.end(), src.begin(), src.end());
    cache_valid_ = false;
  }
  void copy_content(const LidarFactorT& o) {
    const int n = vxba_size(o.h_);
    if (n > 0) {   // device-resident voxels of the source: through the ABI (clusters + cache; fix clusters and weights are not readable back)
      throw std::runtime_error("vxba: copying a factor after its first sweep is not supported (copy it while it only holds pushed voxels)");
    }
    st_clusters_ = o.st_clusters_; st_fix_ = o.st_fix_; st_coe_ = o.st_coe_; st_eigval_ = o.st_eigval_; st_eigvec_ = o.st_eigvec_; st_merged_ = o.st_merged_;
    cache_valid_ = false;
  }
  static PointClusterT unpack_cluster(const double* c) {
    PointClusterT pc;
    pc.P(0, 0) = c[0]; pc.P(0, 1) = pc.P(1, 0) = c[1]; pc.P(0, 2) = pc.P(2, 0) = c[2];
    pc.P(1, 1) = c[3]; pc.P(1, 2) = pc.P(2, 1) = c[4]; pc.P(2, 2) = c[5];
    pc.v[0] = c[6]; pc.v[1] = c[7]; pc.v[2] = c[8];
    pc.N = (int)c[9];
    return pc;
  }
  std::vector<PointClusterT> voxel_clusters(size_t a) const {
    const size_t nd = (size_t)vxba_size(h_);
    std::vector<double> buf((size_t)10 * win_size);
    if (a < nd) check(vxba_read_clusters(h_, (int)a, (int)a + 1, buf.data()));
    else {
      const size_t off = (a - nd) * 10 * win_size;
      if (off + buf.size() > st_clusters_.size()) throw std::out_of_range("vxba: plvec_voxels index");
      std::copy(st_clusters_.begin() + off, st_clusters_.begin() + off + buf.size(), buf.begin());
    }
    std::vector<PointClusterT> out(win_size);
    for (int i = 0; i < win_size; i++) out[i] = unpack_cluster(&buf[(size_t)10 * i]);
    return out;
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
  void sync_cache() const { const_cast<LidarFactorT*>(this)->sync_cache_impl(); }
  void sync_cache_impl() {
    std::lock_guard<std::mutex> lk(mu_);      // OctoTree::margi reads the cache from several threads (multi_margi, voxelslam.cpp:1368-1384)
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
