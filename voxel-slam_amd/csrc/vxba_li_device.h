// Launch interface of the device-resident LiDAR-inertial LM loop (vxba_li_device.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "vxba_kernels.h"

namespace vxli {

constexpr int LI_MAXW = 10;

// Everything the loop keeps between kernels, in device memory.  Flat formats of include/vxba.h (state 24, imu 304 f64).
struct LIState {
  double states[LI_MAXW * 24];           // accepted window states (x_stats)
  double trial[LI_MAXW * 24];            // trial states (x_stats_temp)
  double imus[(LI_MAXW - 1) * 304];      // the factors, with their dbg / dba bookkeeping
  double cov_inv[(LI_MAXW - 1) * 225];   // information matrices, inverted once per call on the host
  double jtj[(LI_MAXW - 1) * 900], gg[(LI_MAXW - 1) * 30], imu_res[LI_MAXW - 1], imu_res_trial[LI_MAXW - 1];
  // the joint system of the last Hessian evaluation, undamped, before the gauge fix, in (pose | velocity-bias) block form
  double Hxx[36 * LI_MAXW * LI_MAXW];    // pose-pose, column-major with leading dimension 6W
  double B[LI_MAXW * 3 * 54];            // B[j][s]: 6x9 block (pose_j, rest_{j-1+s}), column-major
  double Cd[LI_MAXW * 81], Co[LI_MAXW * 81];   // rest-rest: C_jj and C_{j,j+1}, 9x9 column-major
  double g[15 * LI_MAXW];                // gradient (15 per frame)
  double R0[LI_MAXW * 9 * (6 * LI_MAXW + 1)];   // [H_yx | g_y] per velocity-bias block, gauge couplings dropped: the solve's right-hand sides, row-major 9 x (6W+1) per frame
  double dxi[15 * LI_MAXW];
  double u, v, residual1, residual2, q1, imu_coef;
  int calc_hess, done, iter, pad;
  double trace[vxk::LM_MAX_ITER * 8];
  long long dbg[16];                     // development: phase time stamps of the last solve (s_memtime)
};

void launch_li_init(LIState* li, vxk::LMState* lm, int W, double imu_coef, hipStream_t s);
void launch_li_imu(LIState* li, int W, int trial, hipStream_t s);
void launch_li_assemble(LIState* li, const double* d_packed, int W, double* d_hess_out, hipStream_t s);
void launch_li_solve(LIState* li, vxk::LMState* lm, int W, hipStream_t s);
void launch_li_decide(LIState* li, vxk::LMState* lm, const double* d_k2_partial, int nparts, int W, hipStream_t s);

}  // namespace vxli
