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

template <class StateT, class ImuT, class MatXT, class LidarFactorAdapter>
class LI_BA_OptimizerT {
 public:
  int win_size = 0, jac_leng = 0, imu_leng = 0;
  double imu_coef = 1e-4;   // the reference's file-scope `imu_coef` (voxel_map.hpp:446, voxelslam.cpp:822)
  int max_iter = 3;         // hard-coded upstream (:580)

  // voxel_map.hpp:562-653
  void damping_iter(std::vector<StateT>& x_stats, LidarFactorAdapter& voxhess, std::deque<ImuT*>& imus_factor, MatXT* hess) {
    vxba_factor* h = voxhess.handle();   // uploads staged voxels
    win_size = voxhess.win_size;
    jac_leng = 6 * win_size;
    imu_leng = VXBA_LI_DIM * win_size;
    std::vector<double> st((size_t)VXBA_STATE_LEN * win_size), im((size_t)VXBA_IMU_LEN * (win_size > 1 ? win_size - 1 : 0));
    for (int i = 0; i < win_size; i++) pack_state(x_stats[i], &st[(size_t)VXBA_STATE_LEN * i]);
    for (int i = 0; i + 1 < win_size; i++) pack_imu(*imus_factor[i], &im[(size_t)VXBA_IMU_LEN * i]);
    hess->resize(imu_leng, imu_leng);
    if (vxba_li_damping_iter(h, st.data(), im.data(), imu_coef, max_iter, hess->data(), nullptr, nullptr) != VXBA_OK)
      throw std::runtime_error(std::string("vxba: ") + vxba_last_error(h));
    for (int i = 0; i < win_size; i++) unpack_state(&st[(size_t)VXBA_STATE_LEN * i], x_stats[i]);
    for (int i = 0; i + 1 < win_size; i++) unpack_imu_bias_deltas(&im[(size_t)VXBA_IMU_LEN * i], *imus_factor[i]);
  }
};

}  // namespace vxba
