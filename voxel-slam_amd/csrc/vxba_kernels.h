// Launch interface between the C-ABI layer (vxba_capi.hip) and the gfx950 kernels (vxba_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vxk {

constexpr int MAXW = 10;          // VXBA_MAX_WIN: 6W <= 64 accumulator columns
constexpr int K3_BLOCK = 256;     // 4 waves per workgroup
constexpr int DACC = 28;          // per-frame linear accumulators: g(6) Drr(6) Drt(9) Dtt(6) residual(1)

// Poses travel as a kernel argument (W*96 B <= 960 B): uniform scalar loads in K2, one vector load per lane in K3.
struct PoseArg {
  double Rp[12 * MAXW];  // per frame: R column-major (9) | p (3)   -- the C-ABI pose format
};

// Data layout in HBM (all f64, plane stride VS = voxel capacity rounded up to 64):
//   cl     [W][10][VS]  body-frame clusters, frame-major planes: lane <-> voxel reads are contiguous
//   fix    [10][VS]     world-frame fix clusters
//   coe    [VS]
//   eigval [3][VS], eigvec [9][VS] (plane 3*col+row), merged [10][VS]   -- the (lambda, U, pcr_add) cache
//   aux    [2][VS]      s_k = sqrt(2/(lambda_k - lambda_0)), k = 1,2 (derived from eigval, device-private)
struct FactorView {
  double* cl;
  double* fix;
  double* coe;
  double* eigval;
  double* eigvec;
  double* merged;
  double* aux;
  int VS;
  int W;
};

inline int k3_num_tiles(int W) { return (6 * W + 15) / 16; }
inline int k3_num_tile_pairs(int W) { int nt = k3_num_tiles(W); return nt * (nt + 1) / 2; }
// doubles per workgroup partial: MFMA accumulator tiles (register layout) + per-frame linear accumulators
inline size_t k3_partial_len(int W) { return (size_t)k3_num_tile_pairs(W) * 256 + (size_t)W * DACC; }

// K2: residual sweep over voxels [head,end): merge + covariance + eigen-decomposition, writes the cache,
// block partials of sum coe*lambda_0 into d_partial[0..nblocks).  Returns the number of partials.
int launch_k2_residual(const FactorView& fv, const PoseArg& poses, int head, int end, double* d_partial, hipStream_t s);
// Deterministic sum of n partials into d_out[0].
void launch_sum_partials(const double* d_partial, int n, double* d_out, hipStream_t s);
// Derive aux (gap scales) from eigval for voxels [head,end) (after a caller-seeded cache).
void launch_seed_aux(const FactorView& fv, int head, int end, hipStream_t s);

// K3: Hessian/gradient sweep over voxels [head,end) into per-workgroup partials; returns #workgroups.
int k3_grid_blocks(int device_cus);
int launch_k3_hessian(const FactorView& fv, const PoseArg& poses, int head, int end, double* d_partial, int nblocks, hipStream_t s);
// Cross-workgroup reduction + assembly of the packed [Hess (6W)^2 col-major | JacT 6W | residual] buffer.
void launch_k3_finalize(const double* d_partial, int nblocks, int W, double* d_packed, hipStream_t s);

// K1: clusters of n_voxels*W cells from bucketed points (cell = frame*n_voxels + voxel), written to the
// frame-major planes at voxel offset v0.
void launch_k1_build(const double* d_xyz, const int64_t* d_cell_ptr, int n_voxels, int W, const FactorView& fv, int v0, hipStream_t s);

// K4: plane fit of n packed clusters (AoS n*10) -> eig_val n*3, eig_vec n*9 (col-major).
void launch_k4_plane_fit(const double* d_clusters, int64_t n, double* d_eigval, double* d_eigvec, hipStream_t s);

// Layout plumbing between the C-ABI's packed AoS rows and the device planes.
void launch_scatter_clusters(const double* d_src /*[n][W][10]*/, const FactorView& fv, int v0, int n, hipStream_t s);
void launch_gather_clusters(const FactorView& fv, int head, int n, double* d_dst /*[n][W][10]*/, hipStream_t s);
void launch_scatter_rows(const double* d_src /*[n][K]*/, double* planes, int VS, int v0, int n, int K, hipStream_t s);
void launch_gather_rows(const double* planes, int VS, int head, int n, int K, double* d_dst /*[n][K]*/, hipStream_t s);
void launch_fill(double* p, size_t n, double val, hipStream_t s);
void launch_copy_planes(const double* src, int src_vs, double* dst, int dst_vs, int nplanes, int n, hipStream_t s);
void launch_mfma_probe(const double* dA, const double* dB, double* dD, hipStream_t s);
void launch_count_nnz(const FactorView& fv, int V, unsigned long long* d_out, hipStream_t s);

}  // namespace vxk
