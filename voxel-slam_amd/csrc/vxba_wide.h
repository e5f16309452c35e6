// Launch interface of the wide-window sweeps (vxba_wide.hip): win_size up to WIDE_MAXW, sparse incidence.
#pragma once
#include <hip/hip_runtime.h>

#include "vxba_kernels.h"

namespace vxw {

constexpr int WIDE_MAXW = 128;

// Residual sweep (LidarFactor::evaluate_only_residual, voxel_map.hpp:243-279) for any W <= WIDE_MAXW: one lane per voxel, frames
// streamed from the frame-major planes, unobserved (N == 0) entries skipped after one load.  d_poses: W*12 f64 on the device.
// Wave partials of sum coe*lambda0 into d_partial[0 .. ceil((end-head)/64)); returns their number.
int launch_k2_wide(const vxk::FactorView& fv, const double* d_poses, int head, int end, double* d_partial, hipStream_t s);

// Hessian sweep (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241) for any W <= WIDE_MAXW: one wave per voxel, one lane per
// observed (voxel, frame) entry for the rank-3 rows, then all lanes over the (entry pair, 6x6 element) items; contributions are
// added to the packed [Hess | JacT | residual] buffer with hardware f64 atomics (upper triangle), then mirrored.  d_packed is
// zeroed by the launcher.  Summation order is not fixed: results vary in the last bits from run to run.
void launch_k3_wide(const vxk::FactorView& fv, const double* d_poses, int head, int end, double* d_packed, hipStream_t s);

}  // namespace vxw
