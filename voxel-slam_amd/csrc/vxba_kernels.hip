// Hand-written gfx950 (CDNA4, wave64) kernels of the LiDAR BA hot path.
//
//   K1  k1_build_kernel      per-(voxel,frame) cluster accumulation from bucketed points (LDS-staged)
//   K2  k2_residual_kernel   world merge + 3x3 covariance + Jacobi eigensolve + cache write + sum coe*lambda0
//   K3  k3_hessian_kernel    per-entry rank-3 rows (VALU) -> LDS -> f64 MFMA SYRK into the window Hessian,
//                            gradient / block-diagonal terms in registers, deterministic in-block reduction
//       k3_finalize_kernel   cross-workgroup reduction + packed [Hess | JacT | residual] assembly
//   K4  k4_plane_fit_kernel  batched eig(cluster.cov())
//
// Reference loops these replace: tools.hpp:326-331 (K1), voxel_map.hpp:243-279 (K2), voxel_map.hpp:132-241 (K3),
// voxel_map.hpp:1161-1163 / loop_refine.hpp:363-366 (K4), voxel_map.hpp:323-332 (reduction).
// Mathematics: vxba_math.hpp / SURVEY.md Appendix A.  No float atomics anywhere: every reduction is a fixed
// tree, so results are run-to-run bitwise reproducible for a given launch geometry.
#include "vxba_kernels.h"

#include "vxba_math.hpp"

namespace vxk {

typedef double v4d __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------------
// K2 -- residual sweep.  One lane per voxel, frames unrolled: every load is a 512 B contiguous row of a
// frame-major plane, poses are wave-uniform (scalar loads from the kernarg segment), no cross-lane traffic
// until the final residual reduction.
// ------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64) void k2_residual_kernel(FactorView fv, PoseArg poses, int head, int end, double* __restrict__ partial) {
  const int lane = threadIdx.x;
  const int a = head + blockIdx.x * 64 + lane;
  const size_t VS = (size_t)fv.VS;
  double res = 0.0;
  if (a < end) {
    double SP[6], Sv[3], SN;
#pragma unroll
    for (int k = 0; k < 6; k++) SP[k] = fv.fix[k * VS + a];
#pragma unroll
    for (int k = 0; k < 3; k++) Sv[k] = fv.fix[(6 + k) * VS + a];
    SN = fv.fix[9 * VS + a];
#pragma unroll
    for (int i = 0; i < W; i++) {
      const double* clp = fv.cl + (size_t)i * 10 * VS + a;
      double c[10];
#pragma unroll
      for (int k = 0; k < 10; k++) c[k] = clp[k * VS];
      // N == 0 <=> frame i did not observe this voxel (voxel_map.hpp:258): contributes nothing
      const bool obs = c[9] != 0.0;
#pragma unroll
      for (int k = 0; k < 10; k++) c[k] = obs ? c[k] : 0.0;
      double R[9], p[3];
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = poses.Rp[12 * i + 3 * cc + r];
#pragma unroll
      for (int k = 0; k < 3; k++) p[k] = poses.Rp[12 * i + 9 + k];
      vxm::transform_accumulate(c, c + 6, c[9], R, p, SP, Sv, SN);
    }
    double C[6], lam[3], U[9];
    vxm::cluster_cov(SP, Sv, SN, C);
    vxm::eig_sym3(C, lam, U);
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
    fv.aux[a] = s1;
    fv.aux[VS + a] = s2;
    res = fv.coe[a] * lam[0];
  }
  // fixed-tree wave reduction
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) res += __shfl_down(res, off);
  if (lane == 0) partial[blockIdx.x] = res;
}

__global__ __launch_bounds__(1024) void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double red[1024];
  double s = 0.0;
  for (int k = threadIdx.x; k < n; k += 1024) s += partial[k];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

__global__ void seed_aux_kernel(FactorView fv, int head, int end) {
  const int a = head + blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= end) return;
  const size_t VS = (size_t)fv.VS;
  const double lam[3] = {fv.eigval[a], fv.eigval[VS + a], fv.eigval[2 * VS + a]};
  double s1, s2;
  vxm::gap_scales(lam, s1, s2);
  fv.aux[a] = s1;
  fv.aux[VS + a] = s2;
}

// ------------------------------------------------------------------------------------------------
// K3 -- Hessian / gradient sweep.
//
// Work mapping: one lane per (voxel, frame) entry, NV voxels x W frames per wave-batch, the lane's frame is
// fixed for the whole kernel (pose lives in registers; the 27 linear accumulators g/D of that frame stay in
// registers across batches).  Each entry emits three 6-wide row pieces of the per-voxel 3 x 6W matrix B_a
// (vxm::k3_entry) into a wave-private LDS tile [rows][6W]; the wave then reads the tile back in
// v_mfma_f64_16x16x4_f64 operand order (lane l: row 4kk + l/16, column 16c + l%16 -- the SAME register serves
// as A and B operand) and accumulates  S += B^T B  over the upper-triangular 16x16 tile pairs of the 6W x 6W
// window Hessian.  K (the MFMA reduction index) runs over the stacked rows of all voxels, so no per-voxel
// padding is needed.  H = -S + blockdiag(D) is assembled by k3_finalize_kernel.
// No block barrier in the main loop: LDS tiles are wave-private and DS ops of one wave execute in order.
// ------------------------------------------------------------------------------------------------
template <int W>
struct K3Cfg {
  static constexpr int NT = (6 * W + 15) / 16;        // 16-wide column tiles
  static constexpr int NTP = NT * (NT + 1) / 2;       // upper-triangular tile pairs = MFMA accumulators
  static constexpr int NVCAP = (NT <= 2) ? 12 : 8;
  static constexpr int NV = (64 / W) < NVCAP ? (64 / W) : NVCAP;  // voxels per wave-batch
  static constexpr int NACT = NV * W;                 // active lanes
  static constexpr int KSTEPS = (3 * NV + 3) / 4;     // MFMA K-steps per batch (K = 4 rows each)
  static constexpr int ROWS = 4 * KSTEPS;
  static constexpr int NCOL = 16 * NT;
  static constexpr int RS = NCOL + ((NT & 1) ? 32 : 16);  // row stride == 16 (mod 32) doubles: conflict-free ds_read_b64
  static constexpr int WAVE_LDS = ROWS * RS;          // doubles
};

template <int W>
__global__ __launch_bounds__(K3_BLOCK, 1) void k3_hessian_kernel(FactorView fv, PoseArg poses, int head, int end,
                                                                 double* __restrict__ partial) {
  using C = K3Cfg<W>;
  __shared__ double lds[4 * C::WAVE_LDS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  double* ldsw = lds + wave * C::WAVE_LDS;
  for (int k = lane; k < C::WAVE_LDS; k += 64) ldsw[k] = 0.0;  // pad rows / pad columns stay zero forever

  const bool active = lane < C::NACT;
  const int vl = active ? lane / W : 0;
  const int fi = active ? lane % W : 0;
  double R[9], p[3];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int cc = 0; cc < 3; cc++) R[3 * r + cc] = poses.Rp[12 * fi + 3 * cc + r];
#pragma unroll
  for (int k = 0; k < 3; k++) p[k] = poses.Rp[12 * fi + 9 + k];

  v4d acc[C::NTP];
#pragma unroll
  for (int t = 0; t < C::NTP; t++) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  double dacc[DACC];
#pragma unroll
  for (int k = 0; k < DACC; k++) dacc[k] = 0.0;

  const size_t VS = (size_t)fv.VS;
  const int nb = (end - head + C::NV - 1) / C::NV;
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int lrow = lane >> 4, lcol = lane & 15;

  for (int b = gw; b < nb; b += nw) {
    const int a = head + b * C::NV + vl;
    const bool valid = active && a < end;
    double rows[3][6];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int k = 0; k < 6; k++) rows[r][k] = 0.0;
    if (valid) {
      const double* clp = fv.cl + (size_t)fi * 10 * VS + a;
      double c[10];
#pragma unroll
      for (int k = 0; k < 10; k++) c[k] = clp[k * VS];
      const double coe = fv.coe[a];
      if (fi == 0) dacc[27] += coe * fv.eigval[a];  // residual += coe * lambda_0 (voxel_map.hpp:234), once per voxel
      if (c[9] != 0.0) {                             // voxel_map.hpp:178
        vxm::VoxelCache vc;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          vc.u0[k] = fv.eigvec[(size_t)k * VS + a];
          vc.u1[k] = fv.eigvec[(size_t)(3 + k) * VS + a];
          vc.u2[k] = fv.eigvec[(size_t)(6 + k) * VS + a];
        }
        vc.s1 = fv.aux[a];
        vc.s2 = fv.aux[VS + a];
        vc.invN = 1.0 / fv.merged[9 * VS + a];
#pragma unroll
        for (int k = 0; k < 3; k++) vc.vbar[k] = fv.merged[(size_t)(6 + k) * VS + a] * vc.invN;
        vc.coe = coe;
        vxm::k3_entry(c, c + 6, c[9], R, p, vc, rows, dacc);
      }
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 6; k++) ldsw[(3 * vl + r) * C::RS + 6 * fi + k] = rows[r][k];
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int kk = 0; kk < C::KSTEPS; kk++) {
      double x[C::NT];
#pragma unroll
      for (int c = 0; c < C::NT; c++) x[c] = ldsw[(4 * kk + lrow) * C::RS + 16 * c + lcol];
      int t = 0;
#pragma unroll
      for (int I = 0; I < C::NT; I++)
#pragma unroll
        for (int J = I; J < C::NT; J++) {
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[I], x[J], acc[t], 0, 0, 0);
          t++;
        }
    }
    __builtin_amdgcn_wave_barrier();
  }

  // linear accumulators: sum over the NV lanes sharing my frame (fixed order), result valid in lanes < W
  double dsum[DACC];
#pragma unroll
  for (int k = 0; k < DACC; k++) dsum[k] = dacc[k];
#pragma unroll 1
  for (int j = 1; j < C::NV; j++) {
    const int src = (lane + j * W) & 63;
#pragma unroll
    for (int k = 0; k < DACC; k++) dsum[k] += __shfl(dacc[k], src);
  }

  // deterministic in-block reduction over the 4 waves, then one partial per workgroup
  constexpr size_t PLEN = (size_t)C::NTP * 256 + (size_t)W * DACC;
  double* pout = partial + (size_t)blockIdx.x * PLEN;
  __syncthreads();
#pragma unroll
  for (int t = 0; t < C::NTP; t++) {
#pragma unroll
    for (int j = 0; j < 4; j++) lds[wave * 256 + j * 64 + lane] = acc[t][j];
    __syncthreads();
    pout[t * 256 + tid] = ((lds[tid] + lds[256 + tid]) + lds[512 + tid]) + lds[768 + tid];
    __syncthreads();
  }
  if (lane < W) {
#pragma unroll
    for (int k = 0; k < DACC; k++) lds[wave * (W * DACC) + lane * DACC + k] = dsum[k];
  }
  __syncthreads();
  for (int e = tid; e < W * DACC; e += K3_BLOCK)
    pout[C::NTP * 256 + e] = ((lds[e] + lds[W * DACC + e]) + lds[2 * W * DACC + e]) + lds[3 * W * DACC + e];
}

// Cross-workgroup reduction + assembly.  64 outputs per workgroup x 16 partial-slices; every output sums its
// workgroup partials in a fixed order.  Output o < n^2 is Hess(r = o % n, c = o / n) (column-major), then JacT, residual.
__device__ __forceinline__ int sym6_index(int a, int b) { return a == 0 ? b : (a == 1 ? 2 + b : 5); }  // a <= b

template <int W>
__global__ __launch_bounds__(1024) void k3_finalize_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ packed) {
  using C = K3Cfg<W>;
  constexpr int n = 6 * W;
  constexpr int NOUT = n * n + n + 1;
  constexpr size_t PLEN = (size_t)C::NTP * 256 + (size_t)W * DACC;
  __shared__ double red0[16][64];
  __shared__ double red1[16][64];
  const int el = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int o = blockIdx.x * 64 + el;
  int off0 = -1, off1 = -1;  // source offsets inside a partial: MFMA tile element (enters with -), linear element (+)
  if (o < n * n) {
    int r = o % n, c = o / n;
    if (r > c) { int tmp = r; r = c; c = tmp; }  // symmetric; voxel_map.hpp:237-239 mirrors the upper block triangle
    const int I = r >> 4, J = c >> 4;
    const int t = I * C::NT - (I * (I - 1)) / 2 + (J - I);
    const int ri = r & 15, ci = c & 15;
    off0 = t * 256 + (ri >> 2) * 64 + (ri & 3) * 16 + ci;  // f64 MFMA C/D map: row = lane/16 + 4*reg, col = lane%16
    if (r / 6 == c / 6) {
      const int i = r / 6, a = r % 6, b = c % 6;  // a <= b
      int d;
      if (b < 3) d = 6 + sym6_index(a, b);
      else if (a < 3) d = 12 + 3 * a + (b - 3);
      else d = 21 + sym6_index(a - 3, b - 3);
      off1 = C::NTP * 256 + i * DACC + d;
    }
  } else if (o < n * n + n) {
    const int k = o - n * n;
    off1 = C::NTP * 256 + (k / 6) * DACC + (k % 6);
  } else if (o == n * n + n) {
    off1 = C::NTP * 256 + 27;
  }
  double s0 = 0.0, s1 = 0.0;
  if (o < NOUT) {
    for (int b = slice; b < nblocks; b += 16) {
      const double* pb = partial + (size_t)b * PLEN;
      if (off0 >= 0) s0 += pb[off0];
      if (off1 >= 0) s1 += pb[off1];
    }
  }
  red0[slice][el] = s0;
  red1[slice][el] = s1;
  __syncthreads();
  if (slice == 0 && o < NOUT) {
    double t0 = 0.0, t1 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) { t0 += red0[k][el]; t1 += red1[k][el]; }
    packed[o] = t1 - t0;
  }
}

// ------------------------------------------------------------------------------------------------
// K1 -- cluster build.  256 consecutive cells per workgroup; their points are one contiguous range, staged
// through LDS with fully coalesced loads; each lane then folds its own cell's points in input order with
// unfused multiply/add, i.e. exactly PointCluster::push (tools.hpp:326-331) -- bit-identical to the CPU.
// ------------------------------------------------------------------------------------------------
constexpr int K1_CHUNK = 1024;  // points per LDS stage (24 KB)

__global__ __launch_bounds__(256) void k1_build_kernel(const double* __restrict__ xyz, const long long* __restrict__ cell_ptr,
                                                       int n_voxels, int W, FactorView fv, int v0) {
#pragma clang fp contract(off)
  __shared__ double pts[3 * K1_CHUNK];
  const int tid = threadIdx.x;
  const long long ncells = (long long)n_voxels * W;
  const long long c0 = (long long)blockIdx.x * 256;
  const long long c = c0 + tid;
  const long long cend = (c0 + 256 < ncells) ? c0 + 256 : ncells;
  const long long p_begin = cell_ptr[c0], p_end = cell_ptr[cend];
  long long lo = p_end, hi = p_end;
  if (c < ncells) { lo = cell_ptr[c]; hi = cell_ptr[c + 1]; }
  double P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, P5 = 0, vx = 0, vy = 0, vz = 0, N = 0;
  for (long long chunk = p_begin; chunk < p_end; chunk += K1_CHUNK) {
    const int cnt = (int)((p_end - chunk < K1_CHUNK) ? (p_end - chunk) : K1_CHUNK);
    const double* src = xyz + 3 * chunk;
    for (int k = tid; k < 3 * cnt; k += 256) pts[k] = src[k];
    __syncthreads();
    const long long s = lo > chunk ? lo : chunk;
    const long long e = hi < chunk + cnt ? hi : chunk + cnt;
    for (long long q = s; q < e; q++) {
      const int j = 3 * (int)(q - chunk);
      const double x = pts[j], y = pts[j + 1], z = pts[j + 2];
      N += 1.0;
      P0 += x * x; P1 += x * y; P2 += x * z; P3 += y * y; P4 += y * z; P5 += z * z;
      vx += x; vy += y; vz += z;
    }
    __syncthreads();
  }
  if (c < ncells) {
    const int i = (int)(c / n_voxels);
    const int a = v0 + (int)(c % n_voxels);
    const size_t VS = (size_t)fv.VS;
    double* o = fv.cl + (size_t)i * 10 * VS + a;
    o[0] = P0; o[VS] = P1; o[2 * VS] = P2; o[3 * VS] = P3; o[4 * VS] = P4; o[5 * VS] = P5;
    o[6 * VS] = vx; o[7 * VS] = vy; o[8 * VS] = vz; o[9 * VS] = N;
  }
}

// ------------------------------------------------------------------------------------------------
// K4 -- batched plane fit: eig(P/N - c c^T)   (voxel_map.hpp:1161-1163)
// ------------------------------------------------------------------------------------------------
__global__ void k4_plane_fit_kernel(const double* __restrict__ clusters, long long n, double* __restrict__ eigval, double* __restrict__ eigvec) {
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const double* c = clusters + 10 * a;
  double Cm[6], lam[3], U[9];
  vxm::cluster_cov(c, c + 6, c[9], Cm);
  vxm::eig_sym3(Cm, lam, U);
  for (int k = 0; k < 3; k++) eigval[3 * a + k] = lam[k];
  for (int col = 0; col < 3; col++)
    for (int row = 0; row < 3; row++) eigvec[9 * a + 3 * col + row] = U[3 * row + col];
}

// ------------------------------------------------------------------------------------------------
// layout plumbing
// ------------------------------------------------------------------------------------------------
__global__ void scatter_clusters_kernel(const double* __restrict__ src, FactorView fv, int v0, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = fv.W;
  if (t >= (long long)n * W) return;
  const int al = (int)(t % n), i = (int)(t / n);
  const double* s = src + ((size_t)al * W + i) * 10;
  double* o = fv.cl + (size_t)i * 10 * fv.VS + v0 + al;
  for (int k = 0; k < 10; k++) o[(size_t)k * fv.VS] = s[k];
}
__global__ void gather_clusters_kernel(FactorView fv, int head, int n, double* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = fv.W;
  if (t >= (long long)n * W) return;
  const int al = (int)(t % n), i = (int)(t / n);
  double* d = dst + ((size_t)al * W + i) * 10;
  const double* s = fv.cl + (size_t)i * 10 * fv.VS + head + al;
  for (int k = 0; k < 10; k++) d[k] = s[(size_t)k * fv.VS];
}
__global__ void scatter_rows_kernel(const double* __restrict__ src, double* __restrict__ planes, int VS, int v0, int n, int K) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * K) return;
  const int al = (int)(t % n), k = (int)(t / n);
  planes[(size_t)k * VS + v0 + al] = src[(size_t)al * K + k];
}
__global__ void gather_rows_kernel(const double* __restrict__ planes, int VS, int head, int n, int K, double* __restrict__ dst) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * K) return;
  const int al = (int)(t % n), k = (int)(t / n);
  dst[(size_t)al * K + k] = planes[(size_t)k * VS + head + al];
}
__global__ void fill_kernel(double* p, size_t n, double val) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) p[t] = val;
}
__global__ void copy_planes_kernel(const double* __restrict__ src, int src_vs, double* __restrict__ dst, int dst_vs, int nplanes, int n) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)nplanes * n) return;
  const int a = (int)(t % n), k = (int)(t / n);
  dst[(size_t)k * dst_vs + a] = src[(size_t)k * src_vs + a];
}
__global__ void count_nnz_kernel(FactorView fv, int V, unsigned long long* out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long one = 0;
  if (t < (long long)V * fv.W) {
    const int a = (int)(t % V), i = (int)(t / V);
    one = fv.cl[((size_t)i * 10 + 9) * fv.VS + a] != 0.0;
  }
  const unsigned long long cnt = __popcll(__ballot(one != 0));
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, cnt);
}

// Layout probe for v_mfma_f64_16x16x4_f64 (used by a GPU unit test): D(16x16) = A(16x4) B(4x16) with the operand /
// result lane maps K3 relies on -- A[i=l%16][k=l/16], B[k=l/16][j=l%16], D[row = l/16 + 4*reg][col = l%16].
__global__ __launch_bounds__(64) void mfma_probe_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ D) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];
  const double b = B[(l >> 4) * 16 + (l & 15)];
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
#pragma unroll
  for (int j = 0; j < 4; j++) D[((l >> 4) + 4 * j) * 16 + (l & 15)] = acc[j];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_mfma_probe(const double* dA, const double* dB, double* dD, hipStream_t s) {
  mfma_probe_kernel<<<dim3(1), dim3(64), 0, s>>>(dA, dB, dD);
}

#define VXK_DISPATCH_W(W, ...)                                 \
  switch (W) {                                                 \
    case 1: { constexpr int WW = 1; __VA_ARGS__; } break;      \
    case 2: { constexpr int WW = 2; __VA_ARGS__; } break;      \
    case 3: { constexpr int WW = 3; __VA_ARGS__; } break;      \
    case 4: { constexpr int WW = 4; __VA_ARGS__; } break;      \
    case 5: { constexpr int WW = 5; __VA_ARGS__; } break;      \
    case 6: { constexpr int WW = 6; __VA_ARGS__; } break;      \
    case 7: { constexpr int WW = 7; __VA_ARGS__; } break;      \
    case 8: { constexpr int WW = 8; __VA_ARGS__; } break;      \
    case 9: { constexpr int WW = 9; __VA_ARGS__; } break;      \
    case 10: { constexpr int WW = 10; __VA_ARGS__; } break;    \
    default: break;                                            \
  }

int launch_k2_residual(const FactorView& fv, const PoseArg& poses, int head, int end, double* d_partial, hipStream_t s) {
  const int nblocks = (end - head + 63) / 64;
  if (nblocks <= 0) return 0;
  VXK_DISPATCH_W(fv.W, k2_residual_kernel<WW><<<dim3(nblocks), dim3(64), 0, s>>>(fv, poses, head, end, d_partial));
  return nblocks;
}

void launch_sum_partials(const double* d_partial, int n, double* d_out, hipStream_t s) {
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(1024), 0, s, d_partial, n, d_out);
}

void launch_seed_aux(const FactorView& fv, int head, int end, hipStream_t s) {
  if (end <= head) return;
  hipLaunchKernelGGL(seed_aux_kernel, dim3((end - head + 255) / 256), dim3(256), 0, s, fv, head, end);
}

int k3_grid_blocks(int device_cus) { return device_cus; }  // one 4-wave workgroup per CU = one wave per SIMD (the kernel needs > 256 VGPRs)

int launch_k3_hessian(const FactorView& fv, const PoseArg& poses, int head, int end, double* d_partial, int nblocks, hipStream_t s) {
  VXK_DISPATCH_W(fv.W, k3_hessian_kernel<WW><<<dim3(nblocks), dim3(K3_BLOCK), 0, s>>>(fv, poses, head, end, d_partial));
  return nblocks;
}

void launch_k3_finalize(const double* d_partial, int nblocks, int W, double* d_packed, hipStream_t s) {
  const int n = 6 * W, nout = n * n + n + 1;
  VXK_DISPATCH_W(W, k3_finalize_kernel<WW><<<dim3((nout + 63) / 64), dim3(1024), 0, s>>>(d_partial, nblocks, d_packed));
}

void launch_k1_build(const double* d_xyz, const int64_t* d_cell_ptr, int n_voxels, int W, const FactorView& fv, int v0, hipStream_t s) {
  const long long ncells = (long long)n_voxels * W;
  if (ncells <= 0) return;
  hipLaunchKernelGGL(k1_build_kernel, dim3((unsigned)((ncells + 255) / 256)), dim3(256), 0, s, d_xyz, (const long long*)d_cell_ptr, n_voxels, W, fv, v0);
}

void launch_k4_plane_fit(const double* d_clusters, int64_t n, double* d_eigval, double* d_eigvec, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k4_plane_fit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_clusters, (long long)n, d_eigval, d_eigvec);
}

static inline unsigned nblk(long long n, int b) { return (unsigned)((n + b - 1) / b); }

void launch_scatter_clusters(const double* d_src, const FactorView& fv, int v0, int n, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_clusters_kernel, dim3(nblk((long long)n * fv.W, 256)), dim3(256), 0, s, d_src, fv, v0, n);
}
void launch_gather_clusters(const FactorView& fv, int head, int n, double* d_dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_clusters_kernel, dim3(nblk((long long)n * fv.W, 256)), dim3(256), 0, s, fv, head, n, d_dst);
}
void launch_scatter_rows(const double* d_src, double* planes, int VS, int v0, int n, int K, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(nblk((long long)n * K, 256)), dim3(256), 0, s, d_src, planes, VS, v0, n, K);
}
void launch_gather_rows(const double* planes, int VS, int head, int n, int K, double* d_dst, hipStream_t s) {
  if (n <= 0) return;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(nblk((long long)n * K, 256)), dim3(256), 0, s, planes, VS, head, n, K, d_dst);
}
void launch_fill(double* p, size_t n, double val, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(fill_kernel, dim3(nblk((long long)n, 256)), dim3(256), 0, s, p, n, val);
}
void launch_copy_planes(const double* src, int src_vs, double* dst, int dst_vs, int nplanes, int n, hipStream_t s) {
  if (n <= 0 || nplanes <= 0) return;
  hipLaunchKernelGGL(copy_planes_kernel, dim3(nblk((long long)nplanes * n, 256)), dim3(256), 0, s, src, src_vs, dst, dst_vs, nplanes, n);
}
void launch_count_nnz(const FactorView& fv, int V, unsigned long long* d_out, hipStream_t s) {
  if (V <= 0) return;
  hipLaunchKernelGGL(count_nnz_kernel, dim3(nblk((long long)V * fv.W, 256)), dim3(256), 0, s, fv, V, d_out);
}

}  // namespace vxk
