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

// Incidence structure of a wide factor (depends on the clusters only, not on the poses): entries (voxel, frame) and, for every
// 6x6 block of the Hessian, the run of entry pairs that contribute to it.  Device arrays owned by the index.
struct WideIndex {
  int V = 0, nkeys = 0;
  long long nnz = 0, np = 0;
  int* entry_voxel = nullptr;            // [nnz]
  int* entry_frame = nullptr;            // [nnz]
  unsigned int* sei = nullptr;           // [np] first entry of a pair, sorted by block key (frame_i * W + frame_j), voxel order inside
  unsigned int* sej = nullptr;           // [np] second entry
  unsigned int* key_list = nullptr;      // [nkeys] block keys present
  long long* key_ptr = nullptr;          // [nkeys + 1] runs of sei / sej
  double* rowbuf = nullptr;              // [nnz][45] per-entry rows + gradient / block-diagonal terms of the current sweep
};
int build_index(const vxk::FactorView& fv, int V, WideIndex& wi, hipStream_t s, const char** err);   // 0, or -1 with *err set
void free_index(WideIndex& wi);

// Hessian sweep (LidarFactor::acc_evaluate2, voxel_map.hpp:132-241) for any W <= WIDE_MAXW, pair-major: (A) one lane per observed
// (voxel, frame) entry computes its rank-3 rows and gradient / block-diagonal terms, (B) one wave per 6x6 Hessian block sums its
// run of entry pairs in registers and a fixed butterfly, and writes the block and its mirror image.  No atomics: bitwise
// reproducible.  d_partial: scratch for the residual's wave partials (>= ceil((end - head) / 64) doubles).
void launch_k3_wide(const vxk::FactorView& fv, const double* d_poses, const WideIndex& wi, int head, int end, double* d_packed, double* d_partial,
                    hipStream_t s);

// Device-side damped step of the wide LM shell ((H + u D) dxi = -JacT after the gauge fix; dense Cholesky from hipSOLVER, loaded on
// first use).  create returns nullptr if the library cannot be used; step returns non-zero if this step must be taken on the host.
struct DenseSolver;
DenseSolver* wide_solver_create(int n, hipStream_t s);
void wide_solver_free(DenseSolver*& ds);
int wide_solver_step(DenseSolver* ds, const double* d_packed, double u, hipStream_t s, double* dxi, double* q1, double* residual1);

}  // namespace vxw
