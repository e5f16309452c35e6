// Sweeps for WIDE windows (win_size 11 .. 128): the top level of the hierarchical BA runs Lidar_BA_Optimizer on ~100 submap poses
// (voxelslam.cpp:2485-2595, loop_refine.hpp:273-537) where a voxel is seen from a handful of the frames.  The MFMA kernels of
// vxba_kernels.hip are built around 6W <= 64 dense columns; here the incidence is sparse, the Hessian is (6W)^2 = up to 768^2, and
// the per-voxel work is sum over observed PAIRS -- so: same mathematics (vxm::k3_entry rows, H = blockdiag(D) - G^T G), different
// mapping.  Storage is unchanged (frame-major planes, N == 0 marks an unobserved frame).
#include "vxba_wide.h"

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "vxba_math.hpp"

namespace vxw {

using vxk::FactorView;

__global__ __launch_bounds__(64) void k2_wide_kernel(FactorView fv, const double* __restrict__ poses, int head, int end, double* __restrict__ partial) {
  __shared__ double pl[12 * WIDE_MAXW];
  const int lane = threadIdx.x, W = fv.W;
  for (int k = lane; k < 12 * W; k += 64) pl[k] = poses[k];
  __syncthreads();
  const int a = head + blockIdx.x * 64 + lane;
  const size_t VS = (size_t)fv.VS;
  double res = 0.0;
  if (a < end) {
    double SP[6], Sv[3], SN, Up[9];
#pragma unroll
    for (int k = 0; k < 6; k++) SP[k] = fv.fix[k * VS + a];
#pragma unroll
    for (int k = 0; k < 3; k++) Sv[k] = fv.fix[(6 + k) * VS + a];
    SN = fv.fix[9 * VS + a];
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) Up[3 * row + col] = fv.eigvec[(size_t)(3 * col + row) * VS + a];
    for (int i = 0; i < W; i++) {
      const double* c0 = fv.cl + (size_t)i * 10 * VS + a;
      const double n = c0[9 * VS];
      if (n == 0.0) continue;            // frame i did not observe this voxel (voxel_map.hpp:258)
      double c[10];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = c0[(size_t)k * VS];
      c[9] = n;
      double R[9], p[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pl[12 * i + 3 * cc + r];
#pragma unroll
      for (int k = 0; k < 3; k++) p[k] = pl[12 * i + 9 + k];
      vxm::transform_accumulate(c, c + 6, c[9], R, p, SP, Sv, SN);
    }
    double C[6], lam[3], U[9];
    vxm::cluster_cov(SP, Sv, SN, C);
    vxm::eig_sym3_warm(C, Up, lam, U);
#pragma unroll
    for (int k = 0; k < 3; k++) fv.eigval[k * VS + a] = lam[k];
#pragma unroll
    for (int col = 0; col < 3; col++)
#pragma unroll
      for (int row = 0; row < 3; row++) fv.eigvec[(3 * col + row) * VS + a] = U[3 * row + col];
#pragma unroll
    for (int k = 0; k < 6; k++) fv.merged[k * VS + a] = SP[k];
#pragma unroll
    for (int k = 0; k < 3; k++) fv.merged[(6 + k) * VS + a] = Sv[k];
    fv.merged[9 * VS + a] = SN;
    double s1, s2;
    vxm::gap_scales(lam, s1, s2);
    const double coe = fv.coe[a];
    fv.aux[a] = s1;
    fv.aux[VS + a] = s2;
    fv.aux[2 * VS + a] = 1.0 / SN;
    fv.aux[3 * VS + a] = sqrt(coe);
    res = coe * lam[0];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) partial[blockIdx.x] = res;
}

__device__ __forceinline__ int sym6(int a, int b) { return a == 0 ? b : (a == 1 ? 2 + b : 5); }   // a <= b < 3

constexpr int K3W_WAVES = 4;
__global__ __launch_bounds__(64 * K3W_WAVES) void k3_wide_kernel(FactorView fv, const double* __restrict__ poses, int head, int end, double* __restrict__ packed) {
  __shared__ double pl[12 * WIDE_MAXW];
  __shared__ double rows_s[K3W_WAVES][WIDE_MAXW][18];
  __shared__ int frame_s[K3W_WAVES][WIDE_MAXW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, W = fv.W, n = 6 * W;
  for (int k = tid; k < 12 * W; k += blockDim.x) pl[k] = poses[k];
  __syncthreads();
  const int a = head + blockIdx.x * K3W_WAVES + wave;
  if (a >= end) return;                  // whole wave
  const size_t VS = (size_t)fv.VS;
  // the voxel's cached plane: every lane holds a copy (broadcast loads)
  vxm::VoxelCache vc;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    vc.u0[k] = fv.eigvec[(size_t)k * VS + a];
    vc.u1[k] = fv.eigvec[(size_t)(3 + k) * VS + a];
    vc.u2[k] = fv.eigvec[(size_t)(6 + k) * VS + a];
  }
  vc.s1 = fv.aux[a];
  vc.s2 = fv.aux[VS + a];
  vc.invN = fv.aux[2 * VS + a];
  vc.sc = fv.aux[3 * VS + a];
  vc.coe = fv.coe[a];
#pragma unroll
  for (int k = 0; k < 3; k++) vc.vbar[k] = fv.merged[(size_t)(6 + k) * VS + a] * vc.invN;
  if (lane == 0) unsafeAtomicAdd(&packed[(size_t)n * n + n], vc.coe * fv.eigval[a]);   // residual += coe * lambda0 (voxel_map.hpp:234)

  // one lane per observed entry (two rounds cover W <= 128); compacted into LDS in frame order
  int k_total = 0;
  for (int base = 0; base < W; base += 64) {
    const int f = base + lane;
    double nn = 0.0;
    if (f < W) nn = fv.cl[((size_t)f * 10 + 9) * VS + a];
    const bool obs = nn != 0.0;
    const unsigned long long m = __ballot(obs);
    const int slot = k_total + __popcll(m & ((1ull << lane) - 1ull));
    if (obs) {
      double c[10];
#pragma unroll
      for (int k = 0; k < 9; k++) c[k] = fv.cl[((size_t)f * 10 + k) * VS + a];
      c[9] = nn;
      double R[9], p[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = pl[12 * f + 3 * cc + r];
#pragma unroll
      for (int k = 0; k < 3; k++) p[k] = pl[12 * f + 9 + k];
      double rows[3][6], acc[27];
#pragma unroll
      for (int k = 0; k < 27; k++) acc[k] = 0.0;
      vxm::k3_entry(c, c + 6, c[9], R, p, vc, rows, acc);
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) rows_s[wave][slot][6 * r + k] = rows[r][k];
      frame_s[wave][slot] = f;
      // gradient and the block-diagonal part D_i (upper triangle of the 6x6 diagonal block)
      double* J = packed + (size_t)n * n + 6 * f;
#pragma unroll
      for (int d = 0; d < 6; d++) unsafeAtomicAdd(&J[d], acc[d]);
#pragma unroll
      for (int x = 0; x < 3; x++)
#pragma unroll
        for (int y = 0; y < 3; y++) {
          if (x <= y) {
            unsafeAtomicAdd(&packed[(size_t)(6 * f + y) * n + 6 * f + x], acc[6 + sym6(x, y)]);           // rotation-rotation
            unsafeAtomicAdd(&packed[(size_t)(6 * f + 3 + y) * n + 6 * f + 3 + x], acc[21 + sym6(x, y)]);  // translation-translation
          }
          unsafeAtomicAdd(&packed[(size_t)(6 * f + 3 + y) * n + 6 * f + x], acc[12 + 3 * x + y]);         // rotation-translation
        }
    }
    k_total += __popcll(m);
  }
  __builtin_amdgcn_wave_barrier();
  // -G^T G over the observed pairs i <= j (frames ascending, so every element lands in the upper triangle)
  for (int i = 0; i < k_total; i++) {
    const int fi = frame_s[wave][i];
    const double* ri = rows_s[wave][i];
    const int items = (k_total - i) * 36;
    for (int it = lane; it < items; it += 64) {
      const int j = i + it / 36, e = it % 36, x = e / 6, y = e % 6;
      if (j == i && x > y) continue;
      const double* rj = rows_s[wave][j];
      const double v = ri[x] * rj[y] + ri[6 + x] * rj[6 + y] + ri[12 + x] * rj[12 + y];
      unsafeAtomicAdd(&packed[(size_t)(6 * frame_s[wave][j] + y) * n + 6 * fi + x], -v);
    }
  }
}

__global__ void mirror_kernel(double* __restrict__ packed, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * n) return;
  const int r = (int)(t % n), c = (int)(t / n);
  if (r > c) packed[(size_t)c * n + r] = packed[(size_t)r * n + c];   // (voxel_map.hpp:237-239)
}

int launch_k2_wide(const FactorView& fv, const double* d_poses, int head, int end, double* d_partial, hipStream_t s) {
  const int nblocks = (end - head + 63) / 64;
  if (nblocks <= 0) return 0;
  k2_wide_kernel<<<dim3(nblocks), dim3(64), 0, s>>>(fv, d_poses, head, end, d_partial);
  return nblocks;
}

void launch_k3_wide(const FactorView& fv, const double* d_poses, int head, int end, double* d_packed, hipStream_t s) {
  const int n = 6 * fv.W;
  (void)hipMemsetAsync(d_packed, 0, ((size_t)n * n + n + 1) * sizeof(double), s);
  if (end > head) {
    const int nblocks = (end - head + K3W_WAVES - 1) / K3W_WAVES;
    k3_wide_kernel<<<dim3(nblocks), dim3(64 * K3W_WAVES), 0, s>>>(fv, d_poses, head, end, d_packed);
    const long long nn = (long long)n * n;
    mirror_kernel<<<dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, s>>>(d_packed, n);
  }
}

}  // namespace vxw
