// vxba_li_optimizer.hpp -- header-only C++ adapter: the reference's `class LI_BA_Optimizer` (VoxelSLAM/src/voxel_map.hpp:
// 446-655) and the parts of `class IMU_PRE` (preintegration.hpp:11-310) the BA touches, on top of the C ABI (vxba.h).
//
//   call sites replaced:  LI_BA_Optimizer opt_lsv; opt_lsv.damping_iter(x_buf, voxhess, imu_pre_buf, &hess);
//                         voxelslam.cpp:1645-1653 (local mapping), :634 (initialisation)
//
// Template over the caller's own types: StateT = IMUST (fields R, p, v, bg, ba, g), ImuT = IMU_PRE (fields R_delta,
// p_delta, v_delta, bg, ba, R_bg, p_bg, p_ba, v_bg, v_ba, dtime, dbg, dba, dbg_buf, dba_buf, cov), MatXT =
// Eigen::MatrixXd.  Only `operator()(r,c)`, `operator[](i)`, `.data()` and `.resize()` are used, fields are copied one
// by one (never memcpy'd: both structs carry Eigen alignment padding).  The voxel sweeps run on the GPU, the inertial
// half on the calling thread meanwhile -- the same split as upstream's worker threads (voxel_map.hpp:487-499).
#pragma once
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "vxba.h"

namespace vxba {

template <class V3>
inline void put3(const V3& v, double* o) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; }
template <class V3>
inline void get3(const double* o, V3& v) { v[0] = o[0]; v[1] = o[1]; v[2] = o[2]; }
template <class M3>
inline void put33(const M3& m, double* o) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) o[3 * c + r] = m(r, c); }
template <class M3>
inline void get33(const double* o, M3& m) { for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m(r, c) = o[3 * c + r]; }

template <class StateT>
inline void pack_state(const StateT& x, double* s) {
  put33(x.R, s); put3(x.p, s + 9); put3(x.v, s + 12); put3(x.bg, s + 15); put3(x.ba, s + 18); put3(x.g, s + 21);
}
template <class StateT>
inline void unpack_state(const double* s, StateT& x) {   // g is not optimised and stays as it was
  get33(s, x.R); get3(s + 9, x.p); get3(s + 12, x.v); get3(s + 15, x.bg); get3(s + 18, x.ba);
}
template <class ImuT>
inline void pack_imu(const ImuT& f, double* b) {
  put33(f.R_delta, b); put3(f.p_delta, b + 9); put3(f.v_delta, b + 12); put3(f.bg, b + 15); put3(f.ba, b + 18);
  put33(f.R_bg, b + 21); put33(f.p_bg, b + 30); put33(f.p_ba, b + 39); put33(f.v_bg, b + 48); put33(f.v_ba, b + 57);
  b[66] = f.dtime;
  put3(f.dbg, b + 67); put3(f.dba, b + 70); put3(f.dbg_buf, b + 73); put3(f.dba_buf, b + 76);
  for (int c = 0; c < VXBA_LI_DIM; c++) for (int r = 0; r < VXBA_LI_DIM; r++) b[79 + VXBA_LI_DIM * c + r] = f.cov(r, c);
}
template <class ImuT>
inline void unpack_imu_bias_deltas(const double* b, ImuT& f) {   // the only fields the optimizer writes (:608-609, 639-643)
  get3(b + 67, f.dbg); get3(b + 70, f.dba); get3(b + 73, f.dbg_buf); get3(b + 76, f.dba_buf);
}

template <class StateT, class ImuT>
struct LiPack {
  std::vector<double> st, im;
  int W;
  LiPack(const std::vector<StateT>& x_stats, const std::deque<ImuT*>& imus_factor, int win_size) : W(win_size) {
    st.resize((size_t)VXBA_STATE_LEN * W);
    im.resize((size_t)VXBA_IMU_LEN * (W > 1 ? W - 1 : 0));
    for (int i = 0; i < W; i++) pack_state(x_stats[i], &st[(size_t)VXBA_STATE_LEN * i]);
    for (int i = 0; i + 1 < W; i++) pack_imu(*imus_factor[i], &im[(size_t)VXBA_IMU_LEN * i]);
  }
  void unpack(std::vector<StateT>& x_stats, std::deque<ImuT*>& imus_factor, bool with_gravity) {
    for (int i = 0; i < W; i++) {
      unpack_state(&st[(size_t)VXBA_STATE_LEN * i], x_stats[i]);
      if (with_gravity) get3(&st[(size_t)VXBA_STATE_LEN * i + 21], x_stats[i].g);
    }
    for (int i = 0; i + 1 < W; i++) unpack_imu_bias_deltas(&im[(size_t)VXBA_IMU_LEN * i], *imus_factor[i]);
  }
};

inline void li_check(vxba_factor* h, int rc) {
  if (rc != VXBA_OK) throw std::runtime_error(std::string("vxba: ") + vxba_last_error(h));
}

template <class StateT, class ImuT, class MatXT, class VecXT, class LidarFactorAdapter>
class LI_BA_OptimizerT {
 public:
  int win_size = 0, jac_leng = 0, imu_leng = 0;
  double imu_coef = 1e-4;   // the reference's file-scope `imu_coef` (voxel_map.hpp:446, voxelslam.cpp:822)
  int max_iter = 3;         // hard-coded upstream (:580)

  // voxel_map.hpp:455-463
  void hess_plus(MatXT& Hess, VecXT& JacT, MatXT& hs, VecXT& js) { vxba_hess_plus(win_size, Hess.data(), JacT.data(), hs.data(), js.data()); }

  // voxel_map.hpp:465-523: the joint (15W) system; Hess / JacT pre-sized by the caller as upstream
  double divide_thread(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor, MatXT& Hess, VecXT& JacT) {
    vxba_factor* h = voxhess.handle();
    win_size = voxhess.win_size; jac_leng = 6 * win_size; imu_leng = VXBA_LI_DIM * win_size;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    double residual = 0;
    li_check(h, vxba_li_evaluate(h, pk.st.data(), pk.im.data(), imu_coef, Hess.data(), JacT.data(), &residual));
    return residual;
  }
  // voxel_map.hpp:525-560
  double only_residual(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor) {
    vxba_factor* h = voxhess.handle();
    win_size = voxhess.win_size; jac_leng = 6 * win_size; imu_leng = VXBA_LI_DIM * win_size;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    double residual = 0;
    li_check(h, vxba_li_only_residual(h, pk.st.data(), pk.im.data(), imu_coef, &residual));
    voxhess.invalidate_cache();
    return residual;
  }
  // voxel_map.hpp:562-653
  void damping_iter(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor, MatXT* hess) {
    vxba_factor* h = voxhess.handle();   // uploads staged voxels
    win_size = voxhess.win_size;
    jac_leng = 6 * win_size;
    imu_leng = VXBA_LI_DIM * win_size;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    hess->resize(imu_leng, imu_leng);
    li_check(h, vxba_li_damping_iter(h, pk.st.data(), pk.im.data(), imu_coef, max_iter, hess->data(), nullptr, nullptr));
    pk.unpack(x_stats, imus_factor, false);
    voxhess.invalidate_cache();
  }
};

// LI_BA_OptimizerGravity (voxel_map.hpp:659-864): the gravity vector joins the unknowns (15W + 3); every frame's g leaves equal.
template <class StateT, class ImuT, class MatXT, class VecXT, class LidarFactorAdapter>
class LI_BA_OptimizerGravityT {
 public:
  int win_size = 0, jac_leng = 0, imu_leng = 0;
  double imu_coef = 1e-4;

  // voxel_map.hpp:663-671
  void hess_plus(MatXT& Hess, VecXT& JacT, MatXT& hs, VecXT& js) { vxba_hess_plus_gravity(win_size, Hess.data(), JacT.data(), hs.data(), js.data()); }
  // voxel_map.hpp:673-736: the joint (15W + 3) system; Hess / JacT pre-sized by the caller as upstream
  double divide_thread(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor, MatXT& Hess, VecXT& JacT) {
    vxba_factor* h = voxhess.handle();
    win_size = voxhess.win_size; jac_leng = 6 * win_size; imu_leng = VXBA_LI_DIM * win_size + 3;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    double residual = 0;
    li_check(h, vxba_li_evaluate_gravity(h, pk.st.data(), pk.im.data(), imu_coef, Hess.data(), JacT.data(), &residual));
    return residual;
  }
  // voxel_map.hpp:738-773
  double only_residual(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor) {
    vxba_factor* h = voxhess.handle();
    win_size = voxhess.win_size; jac_leng = 6 * win_size; imu_leng = VXBA_LI_DIM * win_size + 3;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    double residual = 0;
    li_check(h, vxba_li_only_residual(h, pk.st.data(), pk.im.data(), imu_coef, &residual));
    voxhess.invalidate_cache();
    return residual;
  }
  // voxel_map.hpp:775-862.  Upstream never resizes *hess (the caller's matrix already has the shape); it is resized here if it has not.
  void damping_iter(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor, std::vector<double>& resis,
                    MatXT* hess, int max_iter = 2) {
    vxba_factor* h = voxhess.handle();
    win_size = voxhess.win_size;
    jac_leng = 6 * win_size;
    imu_leng = VXBA_LI_DIM * win_size + 3;
    LiPack<StateT, ImuT> pk(x_stats, imus_factor, win_size);
    if ((int)hess->rows() != imu_leng || (int)hess->cols() != imu_leng) hess->resize(imu_leng, imu_leng);
    double rs[2] = {0, 0};
    li_check(h, vxba_li_damping_iter_gravity(h, pk.st.data(), pk.im.data(), imu_coef, max_iter, hess->data(), rs, nullptr, nullptr));
    pk.unpack(x_stats, imus_factor, true);
    resis.push_back(rs[0]);
    resis.push_back(rs[1]);
    voxhess.invalidate_cache();
  }
};

}  // namespace vxba
