// Odometry point-to-plane update on the GPU -- self-contained translation unit: device plane map, the
// per-point sweep, and the `vxba_lio_*` entry points of include/vxba.h.
//
// What it replaces (VoxelSLAM/src): `lio_state_estimation` voxelslam.cpp:855-958 with `match` voxel_map.hpp:1335-1392, 1674-1698,
// `var_init` / `calcBodyVar` voxelslam.hpp:163-201, `pvec_update` voxelslam.hpp:203-215.
//
// The reference walks, per point and per EKF iteration, an unordered_map of octree roots and then up to two levels of child
// pointers, on one thread.  Here the map is two flat arrays: an open-addressing table of root-voxel keys, and for every table
// slot the 8^max_layer finest cells of that root, each holding the index of the plane record of the leaf that covers it (a leaf
// above the finest level fills all the cells below it), -1 where the reference would find no child or no plane.  The walk's
// comparisons (float-typed voxel index, `wld > voxel_center` with centres built from a float `quater_length`) are reproduced
// operation by operation, so a point lands on the same leaf as in the reference -- including the ~1e-6 of points whose float
// voxel index rounds across a voxel face -- and so is the per-point node cache (`octos[i]` + `inside`) that the reference keeps
// across the iterations of one call.  One lane per point; 34 running sums (HTH upper triangle, HTz, nnt upper triangle, count)
// are reduce-scattered over the wave -> LDS -> one 272-byte partial per workgroup, stored straight into pinned host memory; the
// caller adds the <= 256 partials in index order (the EKF algebra that consumes them runs there anyway).  No float atomics, no
// device-wide fence: bitwise reproducible.  (A first version finished the sum on the device -- last workgroup to arrive, found via an
// atomic ticket; 391 workgroups each executing __threadfence() cost 3/4 of its 141 us, and even with agent-scope relaxed stores
// instead of fences the ticket + the last workgroup's round trips were 6 of 24 us.  64-lane butterflies of all 34 sums were another 8.)
//
// Bound: HBM in principle (72 B per point + the plane records, which stay in L2), launch/latency in practice -- a 100k-point
// scan is 7 MB.  The 15x15 EKF algebra stays on the host between sweeps (4 sweeps per scan upstream).
#include "vxba_wait.hpp"
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vxba.h"
#include "vxba_imu.hpp"
#include "vxba_math.hpp"
#include "vxba_scratch.hpp"
#include "vxba_internal.h"
#include "vxba_solve.hpp"

namespace vxl {

constexpr int PLANE_LEN = 32;    // f64 per plane record: center 3 | normal 3 | radius | hl | box centre 3 | plane_var upper triangle 21
constexpr int NSUM = 34;         // HTH 21 | HTz 6 | nnt 6 | count
constexpr int SWEEP_OUT = 52;    // HTH 36 col-major | HTz 6 | nnt 9 col-major | match_num
constexpr unsigned long long EMPTY_KEY = ~0ull;
constexpr long long LOC_OFF = 1ll << 20;   // root voxel indices in [-2^20, 2^20)
constexpr int BLOCK = 512;        // 8 waves; one point per lane and pass
constexpr int MAX_GRID = 256;     // one workgroup per CU; larger scans loop inside the lanes (multiple of 8: see the XCD mapping)

struct MapView {
  const unsigned long long* keys;
  const int* cells;
  const double* planes;
  unsigned long long cap_mask;
  int cells_per_root;
  int max_layer;
  double voxel_size;
};

struct SweepArg {
  double R[9];        // column-major
  double p[3];
  double rot_var[9];  // cov.block<3,3>(0,0), column-major
  double tsl_var[9];  // cov.block<3,3>(3,3)
};

// State of one lio_state_estimation call when the EKF algebra runs on the device too (all iterations enqueued up front).
struct LioCtl {
  double state[24], x_prop[24];   // x_curr (in/out) and the propagated state the call started from
  double cov[225], cov_inv[225];  // x_curr.cov (in/out), its inverse at entry
  double G[90];                   // G.block<15,6>(0,0) of the last iteration
  double sweeps[4 * SWEEP_OUT];
  double info[4];                 // ok, iterations, match_num, smallest eigenvalue of nnt
  int rematch_num, iter, done, pad;
};

__host__ __device__ inline unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}
__host__ __device__ inline bool pack_key(long long x, long long y, long long z, unsigned long long& key) {
  const long long a = x + LOC_OFF, b = y + LOC_OFF, c = z + LOC_OFF;
  if ((a | b | c) < 0 || a >= 2 * LOC_OFF || b >= 2 * LOC_OFF || c >= 2 * LOC_OFF) return false;
  key = ((unsigned long long)a << 42) | ((unsigned long long)b << 21) | (unsigned long long)c;
  return true;
}

// a*b rounded, then added / subtracted: what the reference's host build (x86-64 without FMA) computes.  The __fmul_rn / __dmul_rn
// intrinsics are plain operators in this toolchain and get contracted into v_fma after inlining; the pragma clears the contract
// flag on exactly these two instructions and survives inlining (checked in the ISA).
__device__ __forceinline__ float mul_sub_unfused(float c, float a, float b) {
#pragma clang fp contract(off)
  const float p = a * b;
  return c - p;
}
__device__ __forceinline__ double mul_add_unfused(double acc, double a, double b) {
#pragma clang fp contract(off)
  const double p = a * b;
  return acc + p;
}

// the float-typed voxel index of match() (voxel_map.hpp:1678-1685): double division, rounded to float, `-= 1` in float below zero,
// truncated to int64
__device__ inline long long voxel_index(double w, double voxel_size) {
  float loc = (float)(w / voxel_size);
  if (loc < 0.0f) loc = __fsub_rn(loc, 1.0f);
  return (long long)loc;
}

// the leaf test of OctoTree::match (voxel_map.hpp:1340-1365); float-typed quantities kept in float, unfused
__device__ inline bool plane_test(const double* __restrict__ pl, const double w[3], const SweepArg& a, const double pnt[3], const double var[6], double& sigma_d,
                                  double nrm[3], double& resi) {
  const double d0 = w[0] - pl[0], d1 = w[1] - pl[1], d2 = w[2] - pl[2];
  nrm[0] = pl[3]; nrm[1] = pl[4]; nrm[2] = pl[5];
  resi = nrm[0] * d0 + nrm[1] * d1 + nrm[2] * d2;
  const float dis_to_plane = (float)fabs(resi);
  const float dis_to_center = (float)(d0 * d0 + d1 * d1 + d2 * d2);
  const float range_dis = mul_sub_unfused(dis_to_center, dis_to_plane, dis_to_plane);
  const float radius = (float)pl[6];
  if (!(range_dis <= __fmul_rn(9.0f, radius))) return false;
  // J plane_var J^T with J = [d, -n]; the record holds the symmetrised upper triangle
  const double J[6] = {d0, d1, d2, -nrm[0], -nrm[1], -nrm[2]};
  const double* S = pl + 11;
  double sigma_l = 0.0;
  int k = 0;
#pragma unroll
  for (int r = 0; r < 6; r++) {
    double t = 0.0;
#pragma unroll
    for (int c = r; c < 6; c++, k++) t += S[k] * J[c] * (c == r ? 1.0 : 2.0);
    sigma_l += J[r] * t;
  }
  // n^T (R var R^T + phat rot_var phat^T + tsl_var) n  with  q = R^T n,  b = phat^T n = n x pnt
  const double q0 = a.R[0] * nrm[0] + a.R[1] * nrm[1] + a.R[2] * nrm[2];
  const double q1 = a.R[3] * nrm[0] + a.R[4] * nrm[1] + a.R[5] * nrm[2];
  const double q2 = a.R[6] * nrm[0] + a.R[7] * nrm[1] + a.R[8] * nrm[2];
  const double b0 = nrm[1] * pnt[2] - nrm[2] * pnt[1], b1 = nrm[2] * pnt[0] - nrm[0] * pnt[2], b2 = nrm[0] * pnt[1] - nrm[1] * pnt[0];
  sigma_l += q0 * (var[0] * q0 + 2.0 * (var[1] * q1 + var[2] * q2)) + q1 * (var[3] * q1 + 2.0 * var[4] * q2) + q2 * var[5] * q2;
  sigma_l += b0 * (a.rot_var[0] * b0 + a.rot_var[3] * b1 + a.rot_var[6] * b2) + b1 * (a.rot_var[1] * b0 + a.rot_var[4] * b1 + a.rot_var[7] * b2) +
             b2 * (a.rot_var[2] * b0 + a.rot_var[5] * b1 + a.rot_var[8] * b2);
  sigma_l += nrm[0] * (a.tsl_var[0] * nrm[0] + a.tsl_var[3] * nrm[1] + a.tsl_var[6] * nrm[2]) + nrm[1] * (a.tsl_var[1] * nrm[0] + a.tsl_var[4] * nrm[1] + a.tsl_var[7] * nrm[2]) +
             nrm[2] * (a.tsl_var[2] * nrm[0] + a.tsl_var[5] * nrm[1] + a.tsl_var[8] * nrm[2]);
  if (!((double)dis_to_plane < 3.0 * sqrt(sigma_l))) return false;
  sigma_d = sigma_l;
  return true;
}

// match(feat_map, ...) + OctoTree::match's descent (voxel_map.hpp:1366-1386, 1674-1698) on the flattened map: plane index or -1
__device__ inline int map_lookup(const MapView& m, const double w[3]) {
  const long long lx = voxel_index(w[0], m.voxel_size), ly = voxel_index(w[1], m.voxel_size), lz = voxel_index(w[2], m.voxel_size);
  unsigned long long key;
  if (!pack_key(lx, ly, lz, key)) return -1;
  // the descent does not depend on the table, so the finest cell is known before the probe and the cell entry is fetched together
  // with the key it belongs to (one dependent round trip instead of two when the first probe hits, which it mostly does at load <= 1/2)
  double c[3] = {(0.5 + (double)lx) * m.voxel_size, (0.5 + (double)ly) * m.voxel_size, (0.5 + (double)lz) * m.voxel_size};   // cut_voxel :1531-1533
  float ql = (float)(m.voxel_size / 4.0);                                                                                 // :1534
  int cell = 0;
  for (int l = 0; l < m.max_layer; l++) {
    const int x0 = w[0] > c[0], x1 = w[1] > c[1], x2 = w[2] > c[2];
    cell = cell * 8 + 4 * x0 + 2 * x1 + x2;
    c[0] += (double)((float)(2 * x0 - 1) * ql); c[1] += (double)((float)(2 * x1 - 1) * ql); c[2] += (double)((float)(2 * x2 - 1) * ql);   // allocate :1039-1042
    ql = ql / 2;
  }
  unsigned long long slot = mix64(key) & m.cap_mask;
  while (true) {
    const unsigned long long k = m.keys[slot];
    const int rec = m.cells[slot * (unsigned long long)m.cells_per_root + cell];
    if (k == key) return rec;
    if (k == EMPTY_KEY) return -1;
    slot = (slot + 1) & m.cap_mask;
  }
}

// Reduce-scatter of N running sums over the 64 lanes of a wave: at every butterfly step a lane keeps one half of its sums and
// hands the other half to its partner, so 17+9+5+3+2+1 = 37 doubles cross lanes instead of 34*6.  Afterwards lane l holds the wave
// total of column `col` if `real >= 1` (34 of the 64 lanes do).  Fixed data flow: bitwise reproducible.
template <int N, int OFF>
__device__ __forceinline__ double wave_reduce_scatter(const double (&v)[N], int lane, int& col, int& real) {
  if constexpr (OFF == 0) {
    return v[0];
  } else {
    constexpr int M = (N + 1) / 2;
    const bool up = (lane & OFF) != 0;
    double keep[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
      const double lo = v[j];
      const double hi = (M + j < N) ? v[M + j] : 0.0;
      keep[j] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, OFF, 64);
    }
    if (up) { col += M; real -= M; } else { real = real < M ? real : M; }
    return wave_reduce_scatter<M, OFF / 2>(keep, lane, col, real);
  }
}

// One pass of voxelslam.cpp:873-919.  pts: SoA planes [pnt 3 | var upper triangle 6] of stride `n_stride`.
__global__ __launch_bounds__(BLOCK) void lio_sweep_kernel(MapView m, SweepArg a, const double* __restrict__ pts, long long n, long long n_stride, int* __restrict__ cache,
                                                          int use_cache, double* __restrict__ partials,
                                                          int* __restrict__ plane_of_point, double* __restrict__ sigma_of_point, const LioCtl* __restrict__ ctl) {
  __shared__ double red[BLOCK / 64][NSUM];
  if (ctl) {   // device-resident estimation: pose and covariance blocks come from the control block, the loop may already be over
    if (ctl->done) return;
#pragma unroll
    for (int k = 0; k < 9; k++) a.R[k] = ctl->state[k];
#pragma unroll
    for (int k = 0; k < 3; k++) a.p[k] = ctl->state[9 + k];
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
      for (int r = 0; r < 3; r++) { a.rot_var[3 * c + r] = ctl->cov[15 * c + r]; a.tsl_var[3 * c + r] = ctl->cov[15 * (3 + c) + 3 + r]; }
    use_cache = ctl->iter > 0;
  }
  double s[NSUM];
#pragma unroll
  for (int k = 0; k < NSUM; k++) s[k] = 0.0;
  // XCD-aware: workgroups are dealt round-robin to the 8 XCDs, each with its own L2; giving XCD x the x-th contiguous eighth of the
  // scan keeps neighbouring returns -- which hit the same plane records -- behind one L2 instead of spreading every record over all eight
  const unsigned per = gridDim.x / 8;   // the launch rounds the grid up to a multiple of 8
  const unsigned vb = (blockIdx.x % 8) * per + blockIdx.x / 8;
  const unsigned nvb = gridDim.x;
  for (long long i = (long long)vb * BLOCK + threadIdx.x; i < n; i += (long long)nvb * BLOCK) {
    double pnt[3], var[6];
#pragma unroll
    for (int k = 0; k < 3; k++) pnt[k] = pts[k * n_stride + i];
#pragma unroll
    for (int k = 0; k < 6; k++) var[k] = pts[(3 + k) * n_stride + i];
    const double w[3] = {a.R[0] * pnt[0] + a.R[3] * pnt[1] + a.R[6] * pnt[2] + a.p[0], a.R[1] * pnt[0] + a.R[4] * pnt[1] + a.R[7] * pnt[2] + a.p[1],
                         a.R[2] * pnt[0] + a.R[5] * pnt[1] + a.R[8] * pnt[2] + a.p[2]};
    int pid = -1;
    bool cached = false;
    const int prev = use_cache ? cache[i] : -1;
    if (prev >= 0) {   // octos[i] != nullptr && octos[i]->inside(wld)   (voxelslam.cpp:892, voxel_map.hpp:1471-1480)
      const double* pl = m.planes + (size_t)prev * PLANE_LEN;
      const double hl = pl[7];
      cached = w[0] >= pl[8] - hl && w[0] <= pl[8] + hl && w[1] >= pl[9] - hl && w[1] <= pl[9] + hl && w[2] >= pl[10] - hl && w[2] <= pl[10] + hl;
      if (cached) pid = prev;
    }
    if (!cached) pid = map_lookup(m, w);
    bool flag = false;
    double sigma_d = 0.0, nrm[3], resi = 0.0;
    if (pid >= 0) flag = plane_test(m.planes + (size_t)pid * PLANE_LEN, w, a, pnt, var, sigma_d, nrm, resi);
    if (flag) {
      cache[i] = pid;   // `oc = this` only on a match (voxel_map.hpp:1359)
      const double R_inv = 1.0 / (0.0005 + sigma_d);
      const double q0 = a.R[0] * nrm[0] + a.R[1] * nrm[1] + a.R[2] * nrm[2], q1 = a.R[3] * nrm[0] + a.R[4] * nrm[1] + a.R[5] * nrm[2],
                   q2 = a.R[6] * nrm[0] + a.R[7] * nrm[1] + a.R[8] * nrm[2];
      const double jac[6] = {pnt[1] * q2 - pnt[2] * q1, pnt[2] * q0 - pnt[0] * q2, pnt[0] * q1 - pnt[1] * q0, nrm[0], nrm[1], nrm[2]};   // phat R^T n | n
      int k = 0;
#pragma unroll
      for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = r; c < 6; c++, k++) s[k] += R_inv * jac[r] * jac[c];
#pragma unroll
      for (int r = 0; r < 6; r++) s[21 + r] -= R_inv * jac[r] * resi;
      s[27] += nrm[0] * nrm[0]; s[28] += nrm[0] * nrm[1]; s[29] += nrm[0] * nrm[2]; s[30] += nrm[1] * nrm[1]; s[31] += nrm[1] * nrm[2]; s[32] += nrm[2] * nrm[2];
      s[33] += 1.0;
    } else if (!use_cache) {
      cache[i] = -1;
    }
    if (plane_of_point) { plane_of_point[i] = flag ? pid : -1; sigma_of_point[i] = flag ? sigma_d : 0.0; }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    int col = 0, real = NSUM;
    const double t = wave_reduce_scatter<NSUM, 32>(s, lane, col, real);
    if (real >= 1) red[wv][col] = t;
  }
  __syncthreads();
  if (threadIdx.x < NSUM) {
    double t = red[0][threadIdx.x];
#pragma unroll
    for (int q = 1; q < BLOCK / 64; q++) t += red[q][threadIdx.x];
    partials[(size_t)vb * NSUM + threadIdx.x] = t;   // host-mapped: 272 contiguous bytes per workgroup, summed by the caller
  }
}

// The iterated-EKF update between two sweeps (voxelslam.cpp:921-947), one workgroup: sum of the sweep's per-workgroup partials, K_1 =
// (H^T H + cov^-1)^-1 by Gauss-Jordan in LDS (symmetric positive definite: no pivoting), G, the step, x_curr (+)= step, the
// convergence / rematch schedule, and on the last iteration cov = (I - G) cov and the degeneracy test on nnt.
__global__ __launch_bounds__(256) void lio_ekf_kernel(LioCtl* __restrict__ ctl, const double* __restrict__ partials, int grid) {
  constexpr int D = 15, CH = 7, PER = 37;   // 7 chains x 37 partials >= MAX_GRID
  __shared__ double fin[CH][NSUM], tot[NSUM], S[D * D], HTH[36], HTz[6], vec[D], sol[D], Gs[D * 6];
  __shared__ int finish;
  if (ctl->done) return;
  const int tid = threadIdx.x;
  if (tid < CH * NSUM) {
    const int col = tid % NSUM, ch = tid / NSUM;
    double v[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
      const int b = ch + q * CH;
      v[q] = b < grid ? partials[(size_t)b * NSUM + col] : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < PER; q++) t += v[q];
    fin[ch][col] = t;
  }
  __syncthreads();
  if (tid < NSUM) {
    double t = fin[0][tid];
#pragma unroll
    for (int q = 1; q < CH; q++) t += fin[q][tid];
    tot[tid] = t;
  }
  __syncthreads();
  const int it = ctl->iter;
  if (tid < SWEEP_OUT) {   // the sweep record, as the host path assembles it
    int src;
    if (tid < 36) { const int r = tid % 6, c = tid / 6, lo = r < c ? r : c, hi = r < c ? c : r; src = lo * 6 - lo * (lo - 1) / 2 + (hi - lo); }
    else if (tid < 42) src = 21 + (tid - 36);
    else if (tid < 51) { const int r = (tid - 42) % 3, c = (tid - 42) / 3, lo = r < c ? r : c, hi = r < c ? c : r; src = 27 + lo * 3 - lo * (lo - 1) / 2 + (hi - lo); }
    else src = 33;
    const double v = tot[src];
    ctl->sweeps[SWEEP_OUT * it + tid] = v;
    if (tid < 36) HTH[tid] = v;
    else if (tid < 42) HTz[tid - 36] = v;
  }
  if (tid < D * D) S[tid] = ctl->cov_inv[tid];
  if (tid == 255) {   // vec = x_prop - x_curr (IMUST::operator-, tools.hpp:164-173), on a lane the inversion does not use
    double dR[9], w[3];
    vxi::m3_tmul(ctl->state, ctl->x_prop, dR);
    vxi::so3_log(dR, w);
    for (int k = 0; k < 3; k++) vec[k] = w[k];
    for (int k = 0; k < 12; k++) vec[3 + k] = ctl->x_prop[9 + k] - ctl->state[9 + k];
  }
  __syncthreads();
  if (tid < 36) S[D * (tid / 6) + tid % 6] += HTH[tid];
  __syncthreads();
  // Gauss-Jordan inverse of S (15x15) in place, one entry per thread (a one-wave version with four entries per lane and wave
  // barriers was slower: 14.7 instead of 10.4 us for the kernel)
  for (int p = 0; p < D; p++) {
    double nv = 0.0;
    const double ip = vxk::fast_rcp_f64(S[p * D + p]);
    if (tid < D * D) {
      const int r = tid % D, c = tid / D;
      if (r == p && c == p) nv = ip;
      else if (r == p) nv = S[c * D + p] * ip;
      else if (c == p) nv = -S[p * D + r] * ip;
      else nv = S[c * D + r] - S[p * D + r] * S[c * D + p] * ip;
    }
    __syncthreads();
    if (tid < D * D) S[tid] = nv;
    __syncthreads();
  }
  if (tid < D * 6) {   // G.block<15,6>(0,0) = K_1.block<15,6>(0,0) * HTH   (K_1 = S now)
    const int r = tid % D, c = tid / D;
    double g = 0.0;
    for (int k = 0; k < 6; k++) g += S[D * k + r] * HTH[6 * c + k];
    Gs[tid] = g;
    ctl->G[tid] = g;
  }
  __syncthreads();
  if (tid < D) {
    double t = 0.0;
    for (int k = 0; k < 6; k++) t += S[D * k + tid] * HTz[k];
    t += vec[tid];
    for (int k = 0; k < 6; k++) t -= Gs[D * k + tid] * vec[k];
    sol[tid] = t;
  }
  __syncthreads();
  if (tid == 0) {   // x_curr += solution; convergence / rematch schedule (voxelslam.cpp:930-946)
    double E[9], Rn[9];
    vxi::so3_exp(sol, E);
    vxi::m3_mul(ctl->state, E, Rn);
    for (int k = 0; k < 9; k++) ctl->state[k] = Rn[k];
    for (int k = 0; k < 12; k++) ctl->state[9 + k] += sol[3 + k];
    const double rot_add = sqrt(sol[0] * sol[0] + sol[1] * sol[1] + sol[2] * sol[2]);
    const double tra_add = sqrt(sol[3] * sol[3] + sol[4] * sol[4] + sol[5] * sol[5]);
    const bool converged = (rot_add * 57.3 < 0.01) && (tra_add * 100 < 0.015);
    int rematch = ctl->rematch_num;
    if (converged || ((rematch == 0) && (it == 4 - 2))) rematch++;
    ctl->rematch_num = rematch;
    ctl->iter = it + 1;
    finish = (rematch >= 2 || it == 4 - 1) ? 1 : 0;
    if (finish) {
      const double C6[6] = {tot[27], tot[28], tot[29], tot[30], tot[31], tot[32]};
      double lam[3], U[9];
      vxm::eig_sym3(C6, lam, U);
      ctl->info[0] = lam[0] < 14 ? 0.0 : 1.0; ctl->info[1] = it + 1; ctl->info[2] = tot[33]; ctl->info[3] = lam[0];
      ctl->done = 1;
    }
  }
  __syncthreads();
  if (finish) {   // cov = (I - G) cov, G zero beyond its first six columns
    double g = 0.0;
    if (tid < D * D) {
      const int r = tid % D, c = tid / D;
      g = ctl->cov[D * c + r];
      for (int k = 0; k < 6; k++) g -= Gs[D * k + r] * ctl->cov[D * c + k];
    }
    __syncthreads();
    if (tid < D * D) ctl->cov[tid] = g;
  }
}

// var_init (voxelslam.hpp:187-201) with calcBodyVar (:164-185): sensor-frame float xyz -> IMU-frame point + covariance
__global__ void lio_var_init_kernel(const float* __restrict__ xyz, long long n, long long n_stride, SweepArg ext, float range_inc, double dir_var, double* __restrict__ pts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double pb[3] = {(double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]};
  if (pb[2] == 0) pb[2] = 0.0001;
  const double nn = sqrt(pb[0] * pb[0] + pb[1] * pb[1] + pb[2] * pb[2]);
  const float range = (float)nn;
  const float range_var = __fmul_rn(range_inc, range_inc);
  const double d[3] = {pb[0] / nn, pb[1] / nn, pb[2] / nn};
  double b1[3] = {1.0, 1.0, -(d[0] + d[1]) / d[2]};
  const double n1 = sqrt(b1[0] * b1[0] + b1[1] * b1[1] + b1[2] * b1[2]);
  b1[0] /= n1; b1[1] /= n1; b1[2] /= n1;
  double b2[3] = {b1[1] * d[2] - b1[2] * d[1], b1[2] * d[0] - b1[0] * d[2], b1[0] * d[1] - b1[1] * d[0]};
  const double n2 = sqrt(b2[0] * b2[0] + b2[1] * b2[1] + b2[2] * b2[2]);
  b2[0] /= n2; b2[1] /= n2; b2[2] /= n2;
  const double r = (double)range;
  // A = range * hat(d) * [b1 b2]
  const double a1[3] = {r * (d[1] * b1[2] - d[2] * b1[1]), r * (d[2] * b1[0] - d[0] * b1[2]), r * (d[0] * b1[1] - d[1] * b1[0])};
  const double a2[3] = {r * (d[1] * b2[2] - d[2] * b2[1]), r * (d[2] * b2[0] - d[0] * b2[2]), r * (d[0] * b2[1] - d[1] * b2[0])};
  double V[9];
  for (int rr = 0; rr < 3; rr++)
    for (int c = 0; c < 3; c++) V[3 * c + rr] = d[rr] * (double)range_var * d[c] + a1[rr] * dir_var * a1[c] + a2[rr] * dir_var * a2[c];
  // pnt = ext.R * pnt + ext.p;  var = ext.R * var * ext.R^T
  const double* E = ext.R;
  double T[9], W[9];
  for (int rr = 0; rr < 3; rr++)
    for (int c = 0; c < 3; c++) T[3 * c + rr] = E[rr] * V[3 * c] + E[3 + rr] * V[3 * c + 1] + E[6 + rr] * V[3 * c + 2];
  for (int rr = 0; rr < 3; rr++)
    for (int c = 0; c < 3; c++) W[3 * c + rr] = T[rr] * E[c] + T[3 + rr] * E[3 + c] + T[6 + rr] * E[6 + c];
  for (int k = 0; k < 3; k++) pts[k * n_stride + i] = E[k] * pb[0] + E[3 + k] * pb[1] + E[6 + k] * pb[2] + ext.p[k];
  pts[3 * n_stride + i] = W[0]; pts[4 * n_stride + i] = 0.5 * (W[3] + W[1]); pts[5 * n_stride + i] = 0.5 * (W[6] + W[2]);
  pts[6 * n_stride + i] = W[4]; pts[7 * n_stride + i] = 0.5 * (W[7] + W[5]); pts[8 * n_stride + i] = W[8];
}

// caller-provided pointVar arrays (pnt n*3, var n*9 column-major) -> SoA planes
__global__ void lio_scan_pack_kernel(const double* __restrict__ pnt, const double* __restrict__ var9, long long n, long long n_stride, double* __restrict__ pts) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 3; k++) pts[k * n_stride + i] = pnt[3 * i + k];
  const double* V = var9 + 9 * i;
  pts[3 * n_stride + i] = V[0]; pts[4 * n_stride + i] = 0.5 * (V[3] + V[1]); pts[5 * n_stride + i] = 0.5 * (V[6] + V[2]);
  pts[6 * n_stride + i] = V[4]; pts[7 * n_stride + i] = 0.5 * (V[7] + V[5]); pts[8 * n_stride + i] = V[8];
}
__global__ void lio_scan_unpack_kernel(const double* __restrict__ pts, long long n, long long n_stride, double* __restrict__ pnt, double* __restrict__ var9) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = 0; k < 3; k++) pnt[3 * i + k] = pts[k * n_stride + i];
  double* V = var9 + 9 * i;
  const double v0 = pts[3 * n_stride + i], v1 = pts[4 * n_stride + i], v2 = pts[5 * n_stride + i], v3 = pts[6 * n_stride + i], v4 = pts[7 * n_stride + i], v5 = pts[8 * n_stride + i];
  V[0] = v0; V[1] = v1; V[2] = v2; V[3] = v1; V[4] = v3; V[5] = v4; V[6] = v2; V[7] = v4; V[8] = v5;
}

// pvec_update (voxelslam.hpp:203-215): world point + world covariance of every scan point
__global__ void lio_pvec_update_kernel(const double* __restrict__ pts, long long n, long long n_stride, SweepArg a, double* __restrict__ pwld, double* __restrict__ var9) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double pnt[3], v[6];
  for (int k = 0; k < 3; k++) pnt[k] = pts[k * n_stride + i];
  for (int k = 0; k < 6; k++) v[k] = pts[(3 + k) * n_stride + i];
  const double V[9] = {v[0], v[1], v[2], v[1], v[3], v[4], v[2], v[4], v[5]};
  const double H[9] = {0, pnt[2], -pnt[1], -pnt[2], 0, pnt[0], pnt[1], -pnt[0], 0};   // hat(pnt), column-major
  double T[9], O[9];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[3 * c + r] = a.R[r] * V[3 * c] + a.R[3 + r] * V[3 * c + 1] + a.R[6 + r] * V[3 * c + 2];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) O[3 * c + r] = T[r] * a.R[c] + T[3 + r] * a.R[3 + c] + T[6 + r] * a.R[6 + c];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) T[3 * c + r] = H[r] * a.rot_var[3 * c] + H[3 + r] * a.rot_var[3 * c + 1] + H[6 + r] * a.rot_var[3 * c + 2];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) O[3 * c + r] += T[r] * H[c] + T[3 + r] * H[3 + c] + T[6 + r] * H[6 + c] + a.tsl_var[3 * c + r];
  for (int k = 0; k < 9; k++) var9[9 * i + k] = O[k];
  for (int k = 0; k < 3; k++) pwld[3 * i + k] = a.R[k] * pnt[0] + a.R[3 + k] * pnt[1] + a.R[6 + k] * pnt[2] + a.p[k];
}

// ---- plane covariance: the map-side producers of the records above (f2's arithmetic) ----------------------------------------
// cov_add of OctoTree::push (voxel_map.hpp:990-992): sum over a cell's points of Bf_var (:91-106), the 9x9 covariance of the
// cluster's (P upper triangle, v) induced by the point covariance.  One lane per cell, points in input order.
__device__ __forceinline__ void cov_add_point(double (&acc)[81], double x, double y, double z, const double* __restrict__ V /* column-major 3x3 */) {
  const double Bi[6][3] = {{2 * x, 0, 0}, {y, x, 0}, {z, 0, x}, {0, 2 * y, 0}, {0, z, y}, {0, 0, 2 * z}};
  double Biup[6][3];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) Biup[r][k] = Bi[r][0] * V[3 * k] + Bi[r][1] * V[3 * k + 1] + Bi[r][2] * V[3 * k + 2];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 6; k++) acc[9 * k + r] += Biup[r][0] * Bi[k][0] + Biup[r][1] * Bi[k][1] + Biup[r][2] * Bi[k][2];
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) { acc[9 * (6 + k) + r] += Biup[r][k]; acc[9 * r + 6 + k] += Biup[r][k]; }
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int k = 0; k < 3; k++) acc[9 * (6 + k) + 6 + r] += V[3 * k + r];
}

__global__ __launch_bounds__(64) void lio_cov_add_kernel(const double* __restrict__ xyz, const double* __restrict__ var9, const long long* __restrict__ cell_ptr, long long n_cells,
                                   double* __restrict__ cov_add) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  double acc[81];
#pragma unroll
  for (int k = 0; k < 81; k++) acc[k] = 0.0;
  for (long long q = cell_ptr[c]; q < cell_ptr[c + 1]; q++) cov_add_point(acc, xyz[3 * q], xyz[3 * q + 1], xyz[3 * q + 2], var9 + 9 * q);
  double* o = cov_add + 81 * c;
#pragma unroll
  for (int k = 0; k < 81; k++) o[k] = acc[k];
}

// What cut_voxel adds to the leaves a scan touches (OctoTree::push, voxel_map.hpp:969-993), from the world points and covariances
// the last pvec_update left on the device: per leaf the PointCluster of its new points (sequential sums in the order given,
// unfused -- bit-identical to PointCluster::push) and their cov_add.  The host tree's bucketing comes in as (cell_ptr, order):
// leaf c owns scan points order[cell_ptr[c] .. cell_ptr[c+1]).  One lane per leaf.
__global__ __launch_bounds__(64) void lio_leaf_stats_kernel(const double* __restrict__ pwld, const double* __restrict__ var9, const long long* __restrict__ cell_ptr,
                                      const int* __restrict__ order, long long n_cells, double* __restrict__ clusters, double* __restrict__ cov_add) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  double acc[81];
#pragma unroll
  for (int k = 0; k < 81; k++) acc[k] = 0.0;
  double P0 = 0, P1 = 0, P2 = 0, P3 = 0, P4 = 0, P5 = 0, vx = 0, vy = 0, vz = 0, N = 0;
  for (long long q = cell_ptr[c]; q < cell_ptr[c + 1]; q++) {
    const long long i = order[q];
    const double x = pwld[3 * i], y = pwld[3 * i + 1], z = pwld[3 * i + 2];
    N += 1.0;     // products rounded before the add, as the host compiler does for PointCluster::push
    P0 = mul_add_unfused(P0, x, x); P1 = mul_add_unfused(P1, x, y); P2 = mul_add_unfused(P2, x, z);
    P3 = mul_add_unfused(P3, y, y); P4 = mul_add_unfused(P4, y, z); P5 = mul_add_unfused(P5, z, z);
    vx += x; vy += y; vz += z;
    cov_add_point(acc, x, y, z, var9 + 9 * i);
  }
  double* o = clusters + 10 * c;
  o[0] = P0; o[1] = P1; o[2] = P2; o[3] = P3; o[4] = P4; o[5] = P5; o[6] = vx; o[7] = vy; o[8] = vz; o[9] = N;
  double* oc = cov_add + 81 * c;
#pragma unroll
  for (int k = 0; k < 81; k++) oc[k] = acc[k];
}

// OctoTree::plane_update (voxel_map.hpp:1118-1146): centre, normal, radius and the 6x6 covariance of (normal, centre) by first-order
// propagation of cov_add through the eigenvector derivative.  One lane per plane.
__global__ __launch_bounds__(64) void lio_plane_update_kernel(long long n, const double* __restrict__ clusters, const double* __restrict__ eig_val, const double* __restrict__ eig_vec,
                                        const double* __restrict__ cov_add, double* __restrict__ center, double* __restrict__ normal, double* __restrict__ plane_var,
                                        double* __restrict__ radius) {
  const long long a = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n) return;
  const double N = clusters[10 * a + 9];
  const double nv = 1.0 / N;
  const double c[3] = {clusters[10 * a + 6] / N, clusters[10 * a + 7] / N, clusters[10 * a + 8] / N};
  const double* U = eig_vec + 9 * a;   // column k = eigenvector k
  const double* lam = eig_val + 3 * a;
  double u_c[3][9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 9; q++) u_c[r][q] = 0.0;
  const double* ul = U;                // l = 0
#pragma unroll
  for (int k = 1; k < 3; k++) {
    const double* uk = U + 3 * k;
    // ukl = u_k u_l^T ; fkl = [ukl(0,0), ukl(1,0)+ukl(0,1), ukl(2,0)+ukl(0,2), ukl(1,1), ukl(1,2)+ukl(2,1), ukl(2,2) | -(u_k.c u_l + u_l.c u_k)]
    const double kc = uk[0] * c[0] + uk[1] * c[1] + uk[2] * c[2], lc = ul[0] * c[0] + ul[1] * c[1] + ul[2] * c[2];
    const double fkl[9] = {uk[0] * ul[0], uk[1] * ul[0] + uk[0] * ul[1], uk[2] * ul[0] + uk[0] * ul[2], uk[1] * ul[1], uk[1] * ul[2] + uk[2] * ul[1], uk[2] * ul[2],
                           -(kc * ul[0] + lc * uk[0]), -(kc * ul[1] + lc * uk[1]), -(kc * ul[2] + lc * uk[2])};
    const double sc = nv / (lam[0] - lam[k]);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int q = 0; q < 9; q++) u_c[r][q] += sc * uk[r] * fkl[q];
  }
  const double* CA = cov_add + 81 * a;  // column-major 9x9
  double Jc[3][9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 9; q++) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; k++) t += u_c[r][k] * CA[9 * q + k];
      Jc[r][q] = t;
    }
  double* P = plane_var + 36 * a;
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int q = 0; q < 3; q++) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 9; k++) t += Jc[r][k] * u_c[q][k];
      P[6 * q + r] = t;
      const double jn = nv * Jc[r][6 + q];
      P[6 * (3 + q) + r] = jn;
      P[6 * r + 3 + q] = jn;
      P[6 * (3 + q) + 3 + r] = nv * nv * CA[9 * (6 + q) + 6 + r];
    }
#pragma unroll
  for (int k = 0; k < 3; k++) { center[3 * a + k] = c[k]; normal[3 * a + k] = ul[k]; }
  radius[a] = (double)(float)lam[2];
}

// ---- map maintenance ---------------------------------------------------------------------------------------------------
__global__ void lio_fill_u64_kernel(unsigned long long* p, long long n, unsigned long long v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
__global__ void lio_fill_i32_kernel(int* p, long long n, int v) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

__device__ inline unsigned long long table_insert(unsigned long long* keys, unsigned long long cap_mask, unsigned long long key, bool* fresh) {
  unsigned long long slot = mix64(key) & cap_mask;
  while (true) {
    const unsigned long long k = atomicCAS(&keys[slot], EMPTY_KEY, key);
    if (k == EMPTY_KEY) { *fresh = true; return slot; }
    if (k == key) { *fresh = false; return slot; }
    slot = (slot + 1) & cap_mask;
  }
}

// Upsert of n leaves.  A leaf at `layer` covers the 8^(max_layer - layer) finest cells below it.  The plane record of a leaf is
// reused when the same node (root, layer, path) is updated, otherwise a new record is appended.
__global__ void lio_map_update_kernel(long long n, const long long* __restrict__ loc, const int* __restrict__ layer, const int* __restrict__ path, const int* __restrict__ is_plane,
                                      const double* __restrict__ center, const double* __restrict__ normal, const double* __restrict__ plane_var, const double* __restrict__ radius,
                                      unsigned long long* keys, unsigned long long cap_mask, int* cells, int cells_per_root, int max_layer, double voxel_size, double* planes,
                                      long long* plane_tag, int* counters /* [0] roots, [1] planes */) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long key;
  pack_key(loc[3 * i], loc[3 * i + 1], loc[3 * i + 2], key);   // range checked on the host
  bool fresh;
  const unsigned long long slot = table_insert(keys, cap_mask, key, &fresh);
  if (fresh) atomicAdd(&counters[0], 1);
  const int L = layer[i];
  int prefix = 0;
  for (int l = 0; l < L; l++) prefix = prefix * 8 + ((path[i] >> (3 * l)) & 7);
  const int span = 1 << (3 * (max_layer - L));
  int* cl = cells + slot * (unsigned long long)cells_per_root + (size_t)prefix * span;
  const long long tag = ((long long)slot << 16) | ((long long)prefix << 4) | (long long)L;
  const bool plane = is_plane ? is_plane[i] != 0 : true;
  if (!plane) {
    for (int q = 0; q < span; q++) cl[q] = -1;
    return;
  }
  // the entry's record into registers before the first store (round 4): as `pl[k] = center[..]` copies every value was its own load -> wait -> store round trip
  // (the outputs may alias the inputs, for all the compiler knows) -- ~30 of them per thread
  double c3[3], n3[3], Pv[36];
#pragma unroll
  for (int k = 0; k < 3; k++) { c3[k] = center[3 * i + k]; n3[k] = normal[3 * i + k]; }
  const double rad_in = radius[i];
#pragma unroll
  for (int k = 0; k < 36; k++) Pv[k] = plane_var[36 * i + k];
  int rec = cl[0];
  if (rec < 0 || plane_tag[rec] != tag) rec = atomicAdd(&counters[1], 1);
  plane_tag[rec] = tag;
  double* pl = planes + (size_t)rec * PLANE_LEN;
  for (int k = 0; k < 3; k++) { pl[k] = c3[k]; pl[3 + k] = n3[k]; }
  pl[6] = (double)(float)rad_in;
  // the node's box: centre by the reference's own recurrence, half length = 2 * quater_length (voxel_map.hpp:1473)
  double c[3] = {(0.5 + (double)loc[3 * i]) * voxel_size, (0.5 + (double)loc[3 * i + 1]) * voxel_size, (0.5 + (double)loc[3 * i + 2]) * voxel_size};
  float ql = (float)(voxel_size / 4.0);
  for (int l = 0; l < L; l++) {
    const int leafnum = (path[i] >> (3 * l)) & 7;
    const int x0 = (leafnum >> 2) & 1, x1 = (leafnum >> 1) & 1, x2 = leafnum & 1;
    c[0] += (double)((float)(2 * x0 - 1) * ql); c[1] += (double)((float)(2 * x1 - 1) * ql); c[2] += (double)((float)(2 * x2 - 1) * ql);
    ql = ql / 2;
  }
  pl[7] = (double)(ql * 2);
  pl[8] = c[0]; pl[9] = c[1]; pl[10] = c[2];
  const double* P = Pv;
  int k = 11;
#pragma unroll
  for (int r = 0; r < 6; r++)
#pragma unroll
    for (int cc = r; cc < 6; cc++, k++) pl[k] = 0.5 * (P[6 * cc + r] + P[6 * r + cc]);
  for (int q = 0; q < span; q++) cl[q] = rec;
}

__global__ void lio_rehash_kernel(const unsigned long long* __restrict__ old_keys, const int* __restrict__ old_cells, long long old_cap, unsigned long long* keys,
                                  unsigned long long cap_mask, int* cells, int cells_per_root, long long* plane_tag, int n_planes_unused) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= old_cap) return;
  const unsigned long long key = old_keys[i];
  if (key == EMPTY_KEY) return;
  bool fresh;
  const unsigned long long slot = table_insert(keys, cap_mask, key, &fresh);
  const int* src = old_cells + (size_t)i * cells_per_root;
  int* dst = cells + slot * (unsigned long long)cells_per_root;
  for (int q = 0; q < cells_per_root; q++) {
    const int rec = src[q];
    dst[q] = rec;
    if (rec >= 0) plane_tag[rec] = ((long long)slot << 16) | (plane_tag[rec] & 0xffff);   // racy but idempotent: every cell of a record carries the same slot
  }
}

}  // namespace vxl

// ---- host side -----------------------------------------------------------------------------------------------------------
struct vxba_lio {
  int device = 0;
  int opt_device_ekf = 1;   // vxba_lio_set_option(VXBA_LIO_OPT_DEVICE_EKF); initial value may come from VXBA_LIO_DEVICE_EKF
  double voxel_size = 1.0;
  int max_layer = 2;
  int cells_per_root = 64;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // map
  unsigned long long* d_keys = nullptr;
  int* d_cells = nullptr;
  long long cap = 0;
  double* d_planes = nullptr;
  long long* d_plane_tag = nullptr;
  long long plane_cap = 0;
  int* d_counters = nullptr;   // [roots, planes]
  int n_roots = 0, n_planes = 0;
  // scan
  double* d_pts = nullptr;
  long long n_pts = 0, pts_stride = 0, pts_cap = 0;
  int* d_cache = nullptr;
  bool cache_valid = false;
  double* d_world = nullptr;     // world points (3n) + world covariances (9n) of the resident scan, as the last pvec_update left them
  long long world_cap = 0;
  bool world_valid = false;
  char* d_stage = nullptr;       // grow-only device staging for host arrays crossing the boundary (no hipMalloc / hipFree per call)
  size_t stage_cap = 0;
  double* h_partials = nullptr;  // pinned, mapped: one 34-number partial per workgroup lands here (zero-copy stores)
  double* d_partials = nullptr;  // device alias of h_partials
  double h_out[vxl::SWEEP_OUT];  // the sweep's 52 numbers, assembled on the host
  vxl::LioCtl* d_ctl = nullptr;  // device-resident estimation: control block + per-workgroup partials in device memory
  double* d_partials_dev = nullptr;
  std::string err;
  std::recursive_mutex mtx;
};

namespace {

#define LIO_HIP(h, call)                                                                             \
  do {                                                                                               \
    hipError_t e__ = (call);                                                                         \
    if (e__ != hipSuccess) {                                                                         \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e__);                                 \
      return VXBA_ERR_HIP;                                                                           \
    }                                                                                                \
  } while (0)
#define LIO_LOCK(h) std::lock_guard<std::recursive_mutex> lk__((h)->mtx)

int lio_fail(vxba_lio* h, int code, const char* msg) { if (h) h->err = msg; return code; }
inline unsigned grid_for(long long n, int block = 256) { return (unsigned)std::max<long long>(1, (n + block - 1) / block); }

int lio_table_alloc(vxba_lio* h, long long cap, unsigned long long** keys, int** cells) {
  LIO_HIP(h, hipMalloc((void**)keys, (size_t)cap * sizeof(unsigned long long)));
  LIO_HIP(h, hipMalloc((void**)cells, (size_t)cap * h->cells_per_root * sizeof(int)));
  vxl::lio_fill_u64_kernel<<<grid_for(cap), 256, 0, h->stream>>>(*keys, cap, vxl::EMPTY_KEY);
  vxl::lio_fill_i32_kernel<<<grid_for(cap * h->cells_per_root), 256, 0, h->stream>>>(*cells, cap * h->cells_per_root, -1);
  LIO_HIP(h, hipGetLastError());
  return VXBA_OK;
}

// room for `more` further roots / planes
int lio_map_reserve(vxba_lio* h, long long more) {
  long long want = 1024;
  while (want < 2 * ((long long)h->n_roots + more)) want *= 2;
  if (want > h->cap) {
    unsigned long long* nk = nullptr; int* nc = nullptr;
    int rc = lio_table_alloc(h, want, &nk, &nc);
    if (rc != VXBA_OK) return rc;
    if (h->cap) {
      vxl::lio_rehash_kernel<<<grid_for(h->cap), 256, 0, h->stream>>>(h->d_keys, h->d_cells, h->cap, nk, (unsigned long long)want - 1, nc, h->cells_per_root, h->d_plane_tag, h->n_planes);
      LIO_HIP(h, hipGetLastError());
      LIO_HIP(h, hipStreamSynchronize(h->stream));
      LIO_HIP(h, hipFree(h->d_keys)); LIO_HIP(h, hipFree(h->d_cells));
    }
    h->d_keys = nk; h->d_cells = nc; h->cap = want;
  }
  const long long pwant = (long long)h->n_planes + more;
  if (pwant > h->plane_cap) {
    long long ncap = std::max<long long>(4096, h->plane_cap);
    while (ncap < pwant) ncap *= 2;
    double* np = nullptr; long long* nt = nullptr;
    LIO_HIP(h, hipMalloc((void**)&np, (size_t)ncap * vxl::PLANE_LEN * sizeof(double)));
    LIO_HIP(h, hipMalloc((void**)&nt, (size_t)ncap * sizeof(long long)));
    if (h->n_planes) {
      LIO_HIP(h, hipMemcpyAsync(np, h->d_planes, (size_t)h->n_planes * vxl::PLANE_LEN * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
      LIO_HIP(h, hipMemcpyAsync(nt, h->d_plane_tag, (size_t)h->n_planes * sizeof(long long), hipMemcpyDeviceToDevice, h->stream));
      LIO_HIP(h, hipStreamSynchronize(h->stream));
    }
    if (h->d_planes) { LIO_HIP(h, hipFree(h->d_planes)); LIO_HIP(h, hipFree(h->d_plane_tag)); }
    h->d_planes = np; h->d_plane_tag = nt; h->plane_cap = ncap;
  }
  return VXBA_OK;
}

// device staging of at least `bytes` (grow-only; contents are not preserved)
int lio_stage(vxba_lio* h, size_t bytes, char** out) {
  if (bytes > h->stage_cap) {
    if (h->d_stage) { LIO_HIP(h, hipStreamSynchronize(h->stream)); LIO_HIP(h, hipFree(h->d_stage)); }
    h->d_stage = nullptr; h->stage_cap = 0;
    size_t cap = std::max<size_t>(bytes, (size_t)1 << 20);
    cap += cap / 4;
    LIO_HIP(h, hipMalloc((void**)&h->d_stage, cap));
    h->stage_cap = cap;
  }
  *out = h->d_stage;
  return VXBA_OK;
}

int lio_scan_reserve(vxba_lio* h, long long n) {
  if (n > h->pts_cap) {
    if (h->d_pts) { LIO_HIP(h, hipStreamSynchronize(h->stream)); LIO_HIP(h, hipFree(h->d_pts)); LIO_HIP(h, hipFree(h->d_cache)); }
    h->d_pts = nullptr; h->d_cache = nullptr;
    long long cap = std::max<long long>(n, 2 * h->pts_cap);
    cap = (cap + 255) / 256 * 256;
    LIO_HIP(h, hipMalloc((void**)&h->d_pts, (size_t)cap * 9 * sizeof(double)));
    LIO_HIP(h, hipMalloc((void**)&h->d_cache, (size_t)cap * sizeof(int)));
    h->pts_cap = cap;
  }
  h->n_pts = n;
  h->pts_stride = h->pts_cap;
  h->cache_valid = false;
  h->world_valid = false;
  return VXBA_OK;
}

vxl::MapView map_view(const vxba_lio* h) {
  vxl::MapView m;
  m.keys = h->d_keys; m.cells = h->d_cells; m.planes = h->d_planes; m.cap_mask = (unsigned long long)h->cap - 1;
  m.cells_per_root = h->cells_per_root; m.max_layer = h->max_layer; m.voxel_size = h->voxel_size;
  return m;
}

vxl::SweepArg sweep_arg(const double* state, const double* cov225) {
  vxl::SweepArg a;
  std::memcpy(a.R, state, sizeof(double) * 9);
  std::memcpy(a.p, state + 9, sizeof(double) * 3);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) {
      a.rot_var[3 * c + r] = cov225 ? cov225[15 * c + r] : 0.0;
      a.tsl_var[3 * c + r] = cov225 ? cov225[15 * (3 + c) + 3 + r] : 0.0;
    }
  return a;
}

// enqueue one sweep, wait, and assemble its 52 numbers in h->h_out
int lio_sweep(vxba_lio* h, const double* state, const double* cov225, bool reset_cache, int* d_plane_of_point, double* d_sigma_of_point) {
  if (h->n_pts == 0) {
    std::memset(h->h_out, 0, sizeof(double) * vxl::SWEEP_OUT);
    return VXBA_OK;
  }
  if (h->cap == 0) {   // empty map: nothing matches
    int rc = lio_map_reserve(h, 0);
    if (rc != VXBA_OK) return rc;
  }
  const bool use_cache = h->cache_valid && !reset_cache;
  const unsigned grid = std::min<unsigned>((grid_for(h->n_pts, vxl::BLOCK) + 7) / 8 * 8, vxl::MAX_GRID);   // a multiple of 8: one contiguous share of the scan per XCD
  vxl::lio_sweep_kernel<<<grid, vxl::BLOCK, 0, h->stream>>>(map_view(h), sweep_arg(state, cov225), h->d_pts, h->n_pts, h->pts_stride, h->d_cache, use_cache ? 1 : 0, h->d_partials,
                                                            d_plane_of_point, d_sigma_of_point, nullptr);
  LIO_HIP(h, hipGetLastError());
  // the kernel is ~10 us: poll for its completion instead of sleeping on it (hipStreamSynchronize's wake-up costs more than the kernel)
  hipError_t q;
  q = vxwait::stream_wait(h->stream);
  LIO_HIP(h, q);
  h->cache_valid = true;
  // the per-workgroup partials, added in workgroup order (fixed for a given scan size: bitwise reproducible)
  double t[vxl::NSUM];
  for (int k = 0; k < vxl::NSUM; k++) t[k] = 0.0;
  for (unsigned b = 0; b < grid; b++)
    for (int k = 0; k < vxl::NSUM; k++) t[k] += h->h_partials[(size_t)b * vxl::NSUM + k];
  double* o = h->h_out;
  int k = 0;
  for (int r = 0; r < 6; r++)
    for (int c = r; c < 6; c++, k++) { o[6 * c + r] = t[k]; o[6 * r + c] = t[k]; }
  for (int r = 0; r < 6; r++) o[36 + r] = t[21 + r];
  o[42] = t[27]; o[43] = t[28]; o[44] = t[29]; o[45] = t[28]; o[46] = t[30]; o[47] = t[31]; o[48] = t[29]; o[49] = t[31]; o[50] = t[32];
  o[51] = t[33];
  return VXBA_OK;
}

}  // namespace

// completion of what is queued on s by polling: waking up from hipStreamSynchronize costs ~15-25 us, more than the per-scan kernels here
static inline hipError_t lio_poll(hipStream_t s) {
  hipError_t q;
  q = vxwait::stream_wait(s);
  return q;
}

extern "C" {

int vxba_lio_create(double voxel_size, int max_layer, int device, vxba_lio** out) {
  if (!out || !(voxel_size > 0.0) || max_layer < 0 || max_layer > 3) return VXBA_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  vxba_lio* h = new vxba_lio();
  h->device = device; h->voxel_size = voxel_size; h->max_layer = max_layer; h->cells_per_root = 1 << (3 * max_layer);
  { const char* ev = getenv("VXBA_LIO_DEVICE_EKF"); h->opt_device_ekf = !(ev && ev[0] == '0'); }   // initial value only (vxba.h)
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  h->own_stream = e == hipSuccess;
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_counters, 2 * sizeof(int));
  if (e == hipSuccess) e = hipMemset(h->d_counters, 0, 2 * sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->h_partials, (size_t)vxl::MAX_GRID * vxl::NSUM * sizeof(double), hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&h->d_partials, h->h_partials, 0);
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_ctl, sizeof(vxl::LioCtl));
  if (e == hipSuccess) e = hipMalloc((void**)&h->d_partials_dev, (size_t)vxl::MAX_GRID * vxl::NSUM * sizeof(double));
  if (e != hipSuccess) { vxba_lio_destroy(h); return VXBA_ERR_HIP; }
  *out = h;
  return VXBA_OK;
}

int vxba_lio_set_option(vxba_lio* h, int option, int value) {
  if (!h || option != VXBA_LIO_OPT_DEVICE_EKF || (value != 0 && value != 1)) return VXBA_ERR_ARG;
  h->opt_device_ekf = value;
  return VXBA_OK;
}

int vxba_lio_destroy(vxba_lio* h) {
  if (!h) return VXBA_ERR_ARG;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  hipFree(h->d_keys); hipFree(h->d_cells); hipFree(h->d_planes); hipFree(h->d_plane_tag); hipFree(h->d_counters);
  hipFree(h->d_pts); hipFree(h->d_cache); hipFree(h->d_world); hipFree(h->d_stage); hipFree(h->d_ctl); hipFree(h->d_partials_dev);
  if (h->h_partials) hipHostFree(h->h_partials);
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return VXBA_OK;
}

const char* vxba_lio_last_error(const vxba_lio* h) { return h ? h->err.c_str() : "null handle"; }

int vxba_lio_map_clear(vxba_lio* h) {
  if (!h) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  if (h->cap) {
    vxl::lio_fill_u64_kernel<<<grid_for(h->cap), 256, 0, h->stream>>>(h->d_keys, h->cap, vxl::EMPTY_KEY);
    vxl::lio_fill_i32_kernel<<<grid_for(h->cap * h->cells_per_root), 256, 0, h->stream>>>(h->d_cells, h->cap * h->cells_per_root, -1);
    LIO_HIP(h, hipGetLastError());
  }
  LIO_HIP(h, hipMemsetAsync(h->d_counters, 0, 2 * sizeof(int), h->stream));
  LIO_HIP(h, hipStreamSynchronize(h->stream));
  h->n_roots = 0; h->n_planes = 0; h->cache_valid = false;
  return VXBA_OK;
}

int vxba_lio_map_update(vxba_lio* h, int64_t n, const int64_t* loc, const int32_t* layer, const int32_t* path, const int32_t* is_plane, const double* center,
                        const double* normal, const double* plane_var, const double* radius) {
  if (!h || n < 0) return VXBA_ERR_ARG;
  if (n == 0) return VXBA_OK;
  if (!loc || !layer || !path || !center || !normal || !plane_var || !radius) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_map_update: null array");
  LIO_LOCK(h);
  for (int64_t i = 0; i < n; i++) {
    unsigned long long key;
    if (!vxl::pack_key(loc[3 * i], loc[3 * i + 1], loc[3 * i + 2], key)) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_map_update: root voxel index outside [-2^20, 2^20)");
    if (layer[i] < 0 || layer[i] > h->max_layer || (path[i] >> (3 * layer[i])) != 0 || path[i] < 0) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_map_update: layer / path out of range");
  }
  LIO_HIP(h, hipSetDevice(h->device));
  int rc = lio_map_reserve(h, n);
  if (rc != VXBA_OK) return rc;
  // staging: one device buffer for the whole batch
  const size_t b_loc = (size_t)n * 3 * sizeof(int64_t), b_i = (size_t)n * sizeof(int32_t), b_3 = (size_t)n * 3 * sizeof(double), b_36 = (size_t)n * 36 * sizeof(double), b_1 = (size_t)n * sizeof(double);
  const size_t total = b_loc + 3 * ((b_i + 7) / 8 * 8) + 2 * b_3 + b_36 + b_1;
  char* d = nullptr;
  rc = lio_stage(h, total, &d);
  if (rc != VXBA_OK) return rc;
  char* q = d;
  auto put = [&](const void* src, size_t bytes, size_t slot_bytes) -> hipError_t { hipError_t e = hipMemcpyAsync(q, src, bytes, hipMemcpyHostToDevice, h->stream); q += slot_bytes; return e; };
  const long long* d_loc = (const long long*)q; hipError_t e = put(loc, b_loc, b_loc);
  const int* d_layer = (const int*)q; if (e == hipSuccess) e = put(layer, b_i, (b_i + 7) / 8 * 8);
  const int* d_path = (const int*)q; if (e == hipSuccess) e = put(path, b_i, (b_i + 7) / 8 * 8);
  const int* d_isp = is_plane ? (const int*)q : nullptr; if (e == hipSuccess && is_plane) e = put(is_plane, b_i, 0); q += (b_i + 7) / 8 * 8;
  const double* d_center = (const double*)q; if (e == hipSuccess) e = put(center, b_3, b_3);
  const double* d_normal = (const double*)q; if (e == hipSuccess) e = put(normal, b_3, b_3);
  const double* d_pvar = (const double*)q; if (e == hipSuccess) e = put(plane_var, b_36, b_36);
  const double* d_radius = (const double*)q; if (e == hipSuccess) e = put(radius, b_1, b_1);
  if (e == hipSuccess) {
    vxl::lio_map_update_kernel<<<grid_for(n), 256, 0, h->stream>>>(n, d_loc, d_layer, d_path, d_isp, d_center, d_normal, d_pvar, d_radius, h->d_keys, (unsigned long long)h->cap - 1,
                                                                   h->d_cells, h->cells_per_root, h->max_layer, h->voxel_size, h->d_planes, h->d_plane_tag, h->d_counters);
    e = hipGetLastError();
  }
  int cnt[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(cnt, h->d_counters, sizeof cnt, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_map_update: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  h->n_roots = cnt[0]; h->n_planes = cnt[1];
  h->cache_valid = false;   // plane indices held in the per-point cache may be stale
  return VXBA_OK;
}

// ---- links for vxba_map.hip (vxba_internal.h) ----
int vxba_internal_lio_scan_view(vxba_lio* h, const double** d_pts_soa, long long* n, long long* stride, const double** d_world, int* world_valid) {
  if (!h || !d_pts_soa || !n || !stride || !d_world || !world_valid) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  LIO_HIP(h, hipStreamSynchronize(h->stream));
  *d_pts_soa = h->d_pts; *n = h->n_pts; *stride = h->pts_stride; *d_world = h->d_world; *world_valid = h->world_valid ? 1 : 0;
  return VXBA_OK;
}
// vxba_lio_map_update with every array already on the device (loc / layer / path validated by the producer)
int vxba_internal_lio_map_update_device(vxba_lio* h, long long n, const long long* d_loc, const int* d_layer, const int* d_path, const int* d_is_plane, const double* d_center,
                                        const double* d_normal, const double* d_plane_var, const double* d_radius) {
  if (!h || n < 0) return VXBA_ERR_ARG;
  if (n == 0) return VXBA_OK;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  int rc = lio_map_reserve(h, n);
  if (rc != VXBA_OK) return rc;
  vxl::lio_map_update_kernel<<<grid_for(n), 256, 0, h->stream>>>(n, d_loc, d_layer, d_path, d_is_plane, d_center, d_normal, d_plane_var, d_radius, h->d_keys, (unsigned long long)h->cap - 1,
                                                                 h->d_cells, h->cells_per_root, h->max_layer, h->voxel_size, h->d_planes, h->d_plane_tag, h->d_counters);
  hipError_t e = hipGetLastError();
  int cnt[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(cnt, h->d_counters, sizeof cnt, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = std::string("lio_map_update_device: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  h->n_roots = cnt[0]; h->n_planes = cnt[1];
  h->cache_valid = false;
  return VXBA_OK;
}
int vxba_internal_lio_geometry(const vxba_lio* h, double* voxel_size, int* max_layer, int* device) {
  if (!h) return VXBA_ERR_ARG;
  *voxel_size = h->voxel_size; *max_layer = h->max_layer; *device = h->device;
  return VXBA_OK;
}

int vxba_lio_map_size(const vxba_lio* h, int64_t* n_roots, int64_t* n_planes) {
  if (!h) return VXBA_ERR_ARG;
  if (n_roots) *n_roots = h->n_roots;
  if (n_planes) *n_planes = h->n_planes;
  return VXBA_OK;
}

int vxba_lio_scan_set(vxba_lio* h, int64_t n, const double* pnt, const double* var) {
  if (!h || n < 0 || (n > 0 && (!pnt || !var))) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  int rc = lio_scan_reserve(h, n);
  if (rc != VXBA_OK || n == 0) return rc;
  double* d = nullptr;
  rc = lio_stage(h, (size_t)n * 12 * sizeof(double), (char**)&d);
  if (rc != VXBA_OK) return rc;
  hipError_t e = hipMemcpyAsync(d, pnt, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d + 3 * n, var, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    vxl::lio_scan_pack_kernel<<<grid_for(n), 256, 0, h->stream>>>(d, d + 3 * n, n, h->pts_stride, h->d_pts);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_scan_set: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  return VXBA_OK;
}

int vxba_lio_scan_raw(vxba_lio* h, int64_t n, const float* xyz, const double* ext, double dept_err, double beam_err) {
  if (!h || n < 0 || (n > 0 && !xyz)) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  int rc = lio_scan_reserve(h, n);
  if (rc != VXBA_OK || n == 0) return rc;
  vxl::SweepArg e12;
  std::memset(&e12, 0, sizeof e12);
  if (ext) { std::memcpy(e12.R, ext, sizeof(double) * 9); std::memcpy(e12.p, ext + 9, sizeof(double) * 3); }
  else e12.R[0] = e12.R[4] = e12.R[8] = 1.0;
  const float range_inc = (float)dept_err, degree_inc = (float)beam_err;       // calcBodyVar takes them as float (voxelslam.hpp:164)
  const double dir_var = std::pow(std::sin((degree_inc) * 0.017453293), 2);   // pow(sin(DEG2RAD(degree_inc)), 2), PCL's DEG2RAD
  float* d = nullptr;
  rc = lio_stage(h, (size_t)n * 3 * sizeof(float), (char**)&d);
  if (rc != VXBA_OK) return rc;
  hipError_t e = hipMemcpyAsync(d, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    vxl::lio_var_init_kernel<<<grid_for(n), 256, 0, h->stream>>>(d, n, h->pts_stride, e12, range_inc, dir_var, h->d_pts);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = lio_poll(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_scan_raw: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  return VXBA_OK;
}

int64_t vxba_lio_scan_size(const vxba_lio* h) { return h ? h->n_pts : -1; }

int vxba_lio_scan_read(vxba_lio* h, double* pnt, double* var) {
  if (!h || !pnt || !var) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  if (h->n_pts == 0) return VXBA_OK;
  LIO_HIP(h, hipSetDevice(h->device));
  const long long n = h->n_pts;
  double* d = nullptr;
  {
    int rcs = lio_stage(h, (size_t)n * 12 * sizeof(double), (char**)&d);
    if (rcs != VXBA_OK) return rcs;
  }
  vxl::lio_scan_unpack_kernel<<<grid_for(n), 256, 0, h->stream>>>(h->d_pts, n, h->pts_stride, d, d + 3 * n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(pnt, d, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(var, d + 3 * n, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_scan_read: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  return VXBA_OK;
}

int vxba_lio_pvec_update(vxba_lio* h, const double* state, const double* cov, double* pwld, double* var) {
  if (!h || !state || !cov) return VXBA_ERR_ARG;     // pwld == NULL: the result only stays on the device (vxba_map_cut_voxel_lio, vxba_lio_leaf_stats)
  LIO_LOCK(h);
  if (h->n_pts == 0) return VXBA_OK;
  LIO_HIP(h, hipSetDevice(h->device));
  const long long n = h->n_pts;
  if (n > h->world_cap) {
    if (h->d_world) { LIO_HIP(h, hipStreamSynchronize(h->stream)); LIO_HIP(h, hipFree(h->d_world)); }
    h->d_world = nullptr; h->world_cap = 0;
    LIO_HIP(h, hipMalloc((void**)&h->d_world, (size_t)h->pts_cap * 12 * sizeof(double)));
    h->world_cap = h->pts_cap;
  }
  double* d = h->d_world;
  h->world_valid = false;
  vxl::lio_pvec_update_kernel<<<grid_for(n), 256, 0, h->stream>>>(h->d_pts, n, h->pts_stride, sweep_arg(state, cov), d, d + 3 * n);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess && pwld) e = hipMemcpyAsync(pwld, d, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && pwld && var) e = hipMemcpyAsync(var, d + 3 * n, (size_t)n * 9 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = lio_poll(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_pvec_update: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  h->world_valid = true;
  return VXBA_OK;
}

int vxba_lio_leaf_stats(vxba_lio* h, int64_t n_cells, const int64_t* cell_ptr, const int32_t* order, double* clusters, double* cov_add) {
  if (!h || n_cells < 0 || !cell_ptr || !clusters || !cov_add) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  if (n_cells == 0) return VXBA_OK;
  if (!h->world_valid) return lio_fail(h, VXBA_ERR_STATE, "vxba_lio_leaf_stats: no world points on the device (call vxba_lio_pvec_update after loading the scan)");
  const long long m = cell_ptr[n_cells];
  if (cell_ptr[0] != 0 || m < 0 || (m > 0 && !order)) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_leaf_stats: cell_ptr must start at 0 and order must cover cell_ptr[n_cells] entries");
  for (int64_t c = 0; c < n_cells; c++)
    if (cell_ptr[c + 1] < cell_ptr[c]) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_leaf_stats: cell_ptr must be non-decreasing");
  for (long long q = 0; q < m; q++)
    if (order[q] < 0 || order[q] >= h->n_pts) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_leaf_stats: order holds an index outside the resident scan");
  LIO_HIP(h, hipSetDevice(h->device));
  const size_t b_ptr = ((size_t)(n_cells + 1) * 8 + 255) & ~(size_t)255, b_ord = ((size_t)m * 4 + 255) & ~(size_t)255, b_cl = ((size_t)n_cells * 80 + 255) & ~(size_t)255;
  char* d = nullptr;
  {
    int rcs = lio_stage(h, b_ptr + b_ord + b_cl + (size_t)n_cells * 81 * 8, &d);
    if (rcs != VXBA_OK) return rcs;
  }
  long long* d_ptr = (long long*)d;
  int* d_ord = (int*)(d + b_ptr);
  double* d_cl = (double*)(d + b_ptr + b_ord);
  double* d_ca = (double*)(d + b_ptr + b_ord + b_cl);
  hipError_t e = hipMemcpyAsync(d_ptr, cell_ptr, (size_t)(n_cells + 1) * 8, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess && m) e = hipMemcpyAsync(d_ord, order, (size_t)m * 4, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    vxl::lio_leaf_stats_kernel<<<grid_for(n_cells, 64), 64, 0, h->stream>>>(h->d_world, h->d_world + 3 * h->n_pts, d_ptr, d_ord, n_cells, d_cl, d_ca);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(clusters, d_cl, (size_t)n_cells * 80, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(cov_add, d_ca, (size_t)n_cells * 81 * 8, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { h->err = std::string("vxba_lio_leaf_stats: ") + hipGetErrorString(e); return VXBA_ERR_HIP; }
  return VXBA_OK;
}

int vxba_lio_sweep(vxba_lio* h, const double* state, const double* cov, int reset_cache, double* out, int32_t* plane_of_point, double* sigma_of_point) {
  if (!h || !state || !cov || !out) return VXBA_ERR_ARG;
  if ((plane_of_point == nullptr) != (sigma_of_point == nullptr)) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_sweep: plane_of_point and sigma_of_point go together");
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  int* d_pop = nullptr; double* d_sig = nullptr;
  const long long n = h->n_pts;
  if (plane_of_point && n) {
    char* st = nullptr;
    int rcs = lio_stage(h, (size_t)n * (sizeof(double) + sizeof(int)), &st);
    if (rcs != VXBA_OK) return rcs;
    d_sig = (double*)st; d_pop = (int*)(st + (size_t)n * sizeof(double));
  }
  int rc = lio_sweep(h, state, cov, reset_cache != 0, d_pop, d_sig);
  if (rc == VXBA_OK && d_pop) {
    hipError_t e = hipMemcpy(plane_of_point, d_pop, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sigma_of_point, d_sig, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { h->err = std::string("vxba_lio_sweep: ") + hipGetErrorString(e); rc = VXBA_ERR_HIP; }
  }
  if (rc == VXBA_OK) std::memcpy(out, h->h_out, sizeof(double) * vxl::SWEEP_OUT);
  return rc;
}

// lio_state_estimation (voxelslam.cpp:855-958): up to four sweeps, each followed by the 15-dimensional iterated-EKF update on the host.
int vxba_lio_state_estimation(vxba_lio* h, double* state, double* cov, double* info, double* sweeps_out) {
  if (!h || !state || !cov) return VXBA_ERR_ARG;
  LIO_LOCK(h);
  LIO_HIP(h, hipSetDevice(h->device));
  constexpr int D = 15;
  double x_prop[VXBA_STATE_LEN];
  std::memcpy(x_prop, state, sizeof x_prop);
  double cov_inv[D * D], lu[D * D], S[D * D], K1[D * D], G[D * 6];
  int perm[D];
  if (!vxi::dm_inverse(D, cov, cov_inv, lu, perm)) return lio_fail(h, VXBA_ERR_ARG, "vxba_lio_state_estimation: singular state covariance");
  {   // the whole call on the device: sweeps and EKF updates enqueued back to back, one copy each way (VXBA_LIO_DEVICE_EKF=0: host algebra)
    if (h->opt_device_ekf != 0 && h->n_pts > 0) {
      if (h->cap == 0) { int rc = lio_map_reserve(h, 0); if (rc != VXBA_OK) return rc; }
      static thread_local vxl::LioCtl hc;
      std::memset(&hc, 0, sizeof hc);
      std::memcpy(hc.state, state, sizeof hc.state); std::memcpy(hc.x_prop, state, sizeof hc.x_prop);
      std::memcpy(hc.cov, cov, sizeof hc.cov); std::memcpy(hc.cov_inv, cov_inv, sizeof hc.cov_inv);
      LIO_HIP(h, hipMemcpyAsync(h->d_ctl, &hc, sizeof hc, hipMemcpyHostToDevice, h->stream));
      const unsigned grid = std::min<unsigned>((grid_for(h->n_pts, vxl::BLOCK) + 7) / 8 * 8, vxl::MAX_GRID);
      vxl::SweepArg none;
      std::memset(&none, 0, sizeof none);
      for (int it = 0; it < VXBA_LIO_MAX_ITER; it++) {
        vxl::lio_sweep_kernel<<<grid, vxl::BLOCK, 0, h->stream>>>(map_view(h), none, h->d_pts, h->n_pts, h->pts_stride, h->d_cache, 0, h->d_partials_dev, nullptr, nullptr, h->d_ctl);
        vxl::lio_ekf_kernel<<<1, 256, 0, h->stream>>>(h->d_ctl, h->d_partials_dev, (int)grid);
      }
      LIO_HIP(h, hipGetLastError());
      LIO_HIP(h, hipMemcpyAsync(&hc, h->d_ctl, sizeof hc, hipMemcpyDeviceToHost, h->stream));
      hipError_t q;
      q = vxwait::stream_wait(h->stream);
      LIO_HIP(h, q);
      h->cache_valid = true;
      std::memcpy(state, hc.state, sizeof hc.state); std::memcpy(cov, hc.cov, sizeof hc.cov);
      if (info) std::memcpy(info, hc.info, sizeof hc.info);
      if (sweeps_out) std::memcpy(sweeps_out, hc.sweeps, sizeof(double) * vxl::SWEEP_OUT * (int)hc.info[1]);
      return VXBA_OK;
    }
  }
  const int num_max_iter = 4;
  int rematch_num = 0, iterations = 0;
  const double* o = h->h_out;
  for (int iterCount = 0; iterCount < num_max_iter; iterCount++) {
    int rc = lio_sweep(h, state, cov, iterCount == 0, nullptr, nullptr);
    if (rc != VXBA_OK) return rc;
    if (sweeps_out) std::memcpy(sweeps_out + vxl::SWEEP_OUT * iterCount, o, sizeof(double) * vxl::SWEEP_OUT);
    std::memcpy(S, cov_inv, sizeof S);
    for (int c = 0; c < 6; c++)
      for (int r = 0; r < 6; r++) S[D * c + r] += o[6 * c + r];
    if (!vxi::dm_inverse(D, S, K1, lu, perm)) return lio_fail(h, VXBA_ERR_STATE, "vxba_lio_state_estimation: singular information matrix");
    for (int c = 0; c < 6; c++)          // G.block<DIM,6>(0,0) = K_1.block<DIM,6>(0,0) * HTH
      for (int r = 0; r < D; r++) {
        double t = 0;
        for (int k = 0; k < 6; k++) t += K1[D * k + r] * o[6 * c + k];
        G[D * c + r] = t;
      }
    // vec = x_prop - x_curr  (IMUST::operator-, tools.hpp:164-173)
    double vec[D], Rt[9], dR[9], solution[D];
    vxi::m3_tmul(state, x_prop, dR);     // x_curr.R^T * x_prop.R
    (void)Rt;
    vxi::so3_log(dR, vec);
    for (int k = 0; k < 12; k++) vec[3 + k] = x_prop[9 + k] - state[9 + k];
    for (int r = 0; r < D; r++) {
      double t = 0;
      for (int k = 0; k < 6; k++) t += K1[D * k + r] * o[36 + k];
      t += vec[r];
      for (int k = 0; k < 6; k++) t -= G[D * k + r] * vec[k];
      solution[r] = t;
    }
    // x_curr += solution  (IMUST::operator+=, tools.hpp:154-162)
    double E[9], Rn[9];
    vxi::so3_exp(solution, E);
    vxi::m3_mul(state, E, Rn);
    std::memcpy(state, Rn, sizeof Rn);
    for (int k = 0; k < 12; k++) state[9 + k] += solution[3 + k];
    const double rot_add = std::sqrt(solution[0] * solution[0] + solution[1] * solution[1] + solution[2] * solution[2]);
    const double tra_add = std::sqrt(solution[3] * solution[3] + solution[4] * solution[4] + solution[5] * solution[5]);
    const bool converged = (rot_add * 57.3 < 0.01) && (tra_add * 100 < 0.015);
    if (converged || ((rematch_num == 0) && (iterCount == num_max_iter - 2))) rematch_num++;
    iterations = iterCount + 1;
    if (rematch_num >= 2 || (iterCount == num_max_iter - 1)) {
      double nc[D * D];                  // cov = (I - G) * cov, G zero beyond its first six columns
      for (int c = 0; c < D; c++)
        for (int r = 0; r < D; r++) {
          double t = cov[D * c + r];
          for (int k = 0; k < 6; k++) t -= G[D * k + r] * cov[D * c + k];
          nc[D * c + r] = t;
        }
      std::memcpy(cov, nc, sizeof nc);
      break;
    }
  }
  // degeneracy test on the matched normals (voxelslam.cpp:951-957)
  const double C6[6] = {o[42], o[45], o[48], o[46], o[49], o[50]};   // xx xy xz yy yz zz
  double lam[3], U[9];
  vxm::eig_sym3(C6, lam, U);
  if (info) { info[0] = lam[0] < 14 ? 0.0 : 1.0; info[1] = iterations; info[2] = o[51]; info[3] = lam[0]; }
  return VXBA_OK;
}

// Stand-alone, like vxba_plane_fit: cov_add of OctoTree::push for n_cells buckets of world points.
int vxba_cov_add_build(int device, int64_t n_cells, int64_t n_points, const double* xyz_world, const double* var, const int64_t* cell_ptr, double* cov_add) {
  if (n_cells < 0 || n_points < 0 || !cell_ptr || !cov_add || (n_points > 0 && (!xyz_world || !var))) return VXBA_ERR_ARG;
  if (n_cells == 0) return VXBA_OK;
  if (cell_ptr[0] != 0 || cell_ptr[n_cells] != n_points) return VXBA_ERR_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  using vxs::Lease;
  Lease lease(device, Lease::padded((size_t)n_points * 3 * 8) + Lease::padded((size_t)n_points * 9 * 8) + Lease::padded((size_t)(n_cells + 1) * 8) + Lease::padded((size_t)n_cells * 81 * 8));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_xyz = lease.take<double>((size_t)n_points * 3 * 8);
  double* d_var = lease.take<double>((size_t)n_points * 9 * 8);
  long long* d_ptr = lease.take<long long>((size_t)(n_cells + 1) * 8);
  double* d_out = lease.take<double>((size_t)n_cells * 81 * 8);
  hipError_t e = hipSuccess;
  if (e == hipSuccess && n_points) e = hipMemcpy(d_xyz, xyz_world, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess && n_points) e = hipMemcpy(d_var, var, (size_t)n_points * 9 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ptr, cell_ptr, (size_t)(n_cells + 1) * sizeof(long long), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxl::lio_cov_add_kernel<<<grid_for(n_cells, 64), 64>>>(d_xyz, d_var, d_ptr, n_cells, d_out);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(cov_add, d_out, (size_t)n_cells * 81 * sizeof(double), hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

// OctoTree::plane_update (voxel_map.hpp:1118-1146), batched.
int vxba_plane_update(int device, int64_t n, const double* clusters, const double* eig_val, const double* eig_vec, const double* cov_add, double* center, double* normal,
                      double* plane_var, double* radius) {
  if (n < 0 || (n > 0 && (!clusters || !eig_val || !eig_vec || !cov_add || !center || !normal || !plane_var || !radius))) return VXBA_ERR_ARG;
  if (n == 0) return VXBA_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return VXBA_ERR_NODEV;
  if (hipSetDevice(device) != hipSuccess) return VXBA_ERR_HIP;
  const size_t in_len = (size_t)n * (10 + 3 + 9 + 81), out_len = (size_t)n * (3 + 3 + 36 + 1);
  using vxs::Lease;
  Lease lease(device, Lease::padded(in_len * 8) + Lease::padded(out_len * 8));
  if (!lease.ok()) return VXBA_ERR_HIP;
  double* d_in = lease.take<double>(in_len * 8);
  double* d_out = lease.take<double>(out_len * 8);
  hipError_t e = hipSuccess;
  double *d_cl = d_in, *d_ev = d_cl + 10 * n, *d_U = d_ev + 3 * n, *d_ca = d_U + 9 * n;
  double *d_c = d_out, *d_n = d_c + 3 * n, *d_pv = d_n + 3 * n, *d_r = d_pv + 36 * n;
  if (e == hipSuccess) e = hipMemcpy(d_cl, clusters, (size_t)n * 10 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ev, eig_val, (size_t)n * 3 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_U, eig_vec, (size_t)n * 9 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ca, cov_add, (size_t)n * 81 * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    vxl::lio_plane_update_kernel<<<grid_for(n, 64), 64>>>(n, d_cl, d_ev, d_U, d_ca, d_c, d_n, d_pv, d_r);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(center, d_c, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(normal, d_n, (size_t)n * 3 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(plane_var, d_pv, (size_t)n * 36 * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(radius, d_r, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
  return e == hipSuccess ? VXBA_OK : VXBA_ERR_HIP;
}

}  // extern "C"
