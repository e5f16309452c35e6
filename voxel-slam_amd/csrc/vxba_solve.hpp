// Device-side building blocks of the damped solves (included by vxba_kernels.hip): the one-wave dense
// elimination with look-ahead, lane broadcasts, the fast reciprocal, R <- R Exp(dphi).
#pragma once
#include <hip/hip_runtime.h>

namespace vxk {

__device__ __forceinline__ void lm_right_multiply_exp(const double* Rin, const double* dphi, double* Rout) {
  // R <- R Exp(dphi), column-major 3x3; Rodrigues with the reference's 1e-11 cut-off (tools.hpp:51-66)
  const double th2 = dphi[0] * dphi[0] + dphi[1] * dphi[1] + dphi[2] * dphi[2];
  const double th = sqrt(th2);
  double E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (th >= 1e-11 && th2 < 0.0625) {
    // an LM step: |dphi| < 0.25 rad.  E = I + A K + B K^2 with K = hat(dphi) (not normalised), A = sin(th) / th, B = (1 - cos th) / th^2
    // from their series in th^2 (eight terms: the first neglected one is below 1e-24) -- the same matrix as the branch below to
    // round-off, without the library sincos (~150 instructions on one wave, on the critical path between solve and residual sweep).
    double A = -1.0 / 1307674368000.0, Bc = -1.0 / 20922789888000.0;   // -1/15!, -1/16!
    A = fma(A, th2, 1.0 / 6227020800.0);   Bc = fma(Bc, th2, 1.0 / 87178291200.0);    // +1/13!, +1/14!
    A = fma(A, th2, -1.0 / 39916800.0);    Bc = fma(Bc, th2, -1.0 / 479001600.0);     // -1/11!, -1/12!
    A = fma(A, th2, 1.0 / 362880.0);       Bc = fma(Bc, th2, 1.0 / 3628800.0);        // +1/9!,  +1/10!
    A = fma(A, th2, -1.0 / 5040.0);        Bc = fma(Bc, th2, -1.0 / 40320.0);         // -1/7!,  -1/8!
    A = fma(A, th2, 1.0 / 120.0);          Bc = fma(Bc, th2, 1.0 / 720.0);            // +1/5!,  +1/6!
    A = fma(A, th2, -1.0 / 6.0);           Bc = fma(Bc, th2, -1.0 / 24.0);            // -1/3!,  -1/4!
    A = fma(A, th2, 1.0);                  Bc = fma(Bc, th2, 0.5);
    const double k0 = dphi[0], k1 = dphi[1], k2 = dphi[2];
    E[0] = 1.0 + Bc * (k0 * k0 - th2); E[1] = -A * k2 + Bc * k0 * k1;      E[2] = A * k1 + Bc * k0 * k2;
    E[3] = A * k2 + Bc * k0 * k1;      E[4] = 1.0 + Bc * (k1 * k1 - th2); E[5] = -A * k0 + Bc * k1 * k2;
    E[6] = -A * k1 + Bc * k0 * k2;     E[7] = A * k0 + Bc * k1 * k2;      E[8] = 1.0 + Bc * (k2 * k2 - th2);
  } else if (th >= 1e-11) {
    const double ith = 1.0 / th;
    const double k0 = dphi[0] * ith, k1 = dphi[1] * ith, k2 = dphi[2] * ith;
    double sn, cs;
    sincos(th, &sn, &cs);
    const double c1 = 1.0 - cs;
    // E = I + sin K + (1 - cos) K^2,  K = hat(k):  K^2 = k k^T - I (|k| = 1)
    E[0] = 1.0 + c1 * (k0 * k0 - 1.0); E[1] = -sn * k2 + c1 * k0 * k1;      E[2] = sn * k1 + c1 * k0 * k2;
    E[3] = sn * k2 + c1 * k0 * k1;      E[4] = 1.0 + c1 * (k1 * k1 - 1.0); E[5] = -sn * k0 + c1 * k1 * k2;
    E[6] = -sn * k1 + c1 * k0 * k2;     E[7] = sn * k0 + c1 * k1 * k2;      E[8] = 1.0 + c1 * (k2 * k2 - 1.0);
  }
  double out[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) out[3 * c + r] = Rin[r] * E[c] + Rin[3 + r] * E[3 + c] + Rin[6 + r] * E[6 + c];
#pragma unroll
  for (int q = 0; q < 9; q++) Rout[q] = out[q];
}

__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
// 1/d to fp64 round-off: hardware estimate + two Newton steps (the IEEE division sequence is ~3x longer and sits on
// the critical path of every elimination step)
__device__ __forceinline__ double fast_rcp_f64(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// Reciprocal of an elimination pivot with Eigen's rule for null pivots (LDLT::solve zeroes the components whose pivot is not
// above 1/highest()): a frame without any observation in the factor has an all-zero block row, damping u*diag(H) does not lift
// it, and the reference then returns dxi = 0 for that frame while the rest of the window moves.  With inv = 0 the step eliminates
// nothing (multipliers 0) and back substitution yields x = 0 for the row -- the same answer, instead of inf / NaN poses.
__device__ __forceinline__ double pivot_rcp_f64(double d) {
  const double r = fast_rcp_f64(d);
  return fabs(d) > 1e-300 ? r : 0.0;
}

constexpr int LM_PRE = 12;   // pivot-column values fetched one step ahead
template <int K, int N>
struct LmElim {
  // Forward elimination step K with one step of look-ahead.  On entry the (final) column K of the trailing matrix already
  // sits in LDS buffer K & 1, its first LM_PRE entries below the diagonal are in `pre`, and 1/pivot and the pivot's
  // right-hand side are known.  The step updates column K+1 FIRST, publishes it (other LDS buffer), fetches the head of
  // it and starts the next reciprocal -- all of which then complete in the shadow of the remaining rank-1 update,
  // instead of costing an LDS round trip + a division chain per step on the critical path (54 dependent steps).
  // Measured and rejected on top of this: masking rows with exec (real branches) instead of selects, 1/pivot and the
  // solution through LDS, damping added at the pivot read -- 1100 fewer instructions, but 31k instead of 26k cycles: the
  // wave is bound by the LDS queue order and dependent latencies, not by instruction count or cold instruction fetch
  // (a second pass over the same code with a warm I-cache is only 10 % faster).
  static __device__ __forceinline__ void forward(double (&A)[N > 6 ? N : 7], double& b, double& my_invd, double* colbuf, int lane, bool row_ok,
                                                 double invd, double bk, double (&pre)[LM_PRE]) {
    if constexpr (K < N) {
      constexpr int M = N - K - 1;            // columns j = K+1 .. N-1 take the rank-1 update
      const double* cur = colbuf + 64 * (K & 1);
      double* nxt = colbuf + 64 * ((K + 1) & 1);
      double col[M > LM_PRE ? M - LM_PRE : 1];
#pragma unroll
      for (int j = LM_PRE; j < M; j++) col[j - LM_PRE] = cur[K + 1 + j];   // the tail of column K: broadcast reads, back to back
      __builtin_amdgcn_sched_barrier(0);
      my_invd = (lane == K) ? invd : my_invd;
      const double l = (lane > K && row_ok) ? A[K] * invd : 0.0;
      b -= l * bk;
      double invd_n = 0.0, bk_n = 0.0;
      double pre_n[LM_PRE];
#pragma unroll
      for (int j = 0; j < LM_PRE; j++) pre_n[j] = 0.0;
      if constexpr (M > 0) {
        A[K + 1] -= l * pre[0];
        nxt[lane] = A[K + 1];                 // lane j publishes A(j,K+1); symmetric, so this is also row K+1
        const double d_n = readlane_f64(A[K + 1], K + 1);
        bk_n = readlane_f64(b, K + 1);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < LM_PRE && j < M - 1; j++) pre_n[j] = nxt[K + 2 + j];
        __builtin_amdgcn_sched_barrier(0);
        invd_n = pivot_rcp_f64(d_n);
#pragma unroll
        for (int j = 1; j < LM_PRE && j < M; j++) A[K + 1 + j] -= l * pre[j];
#pragma unroll
        for (int j = LM_PRE; j < M; j++) A[K + 1 + j] -= l * col[j - LM_PRE];
      }
      __builtin_amdgcn_sched_barrier(0);
      LmElim<K + 1, N>::forward(A, b, my_invd, colbuf, lane, row_ok, invd_n, bk_n, pre_n);
    }
  }
  static __device__ __forceinline__ void backward(double (&A)[N > 6 ? N : 7], double& b, const double my_invd, double& x, double* xs, int lane) {
    if constexpr (K >= 6) {
      const double xk = readlane_f64(b * my_invd, K);   // x_K = y_K / pivot_K, broadcast from lane K
      x = (lane == K) ? xk : x;
      b -= (lane < K) ? A[K] * xk : 0.0;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
      LmElim<K - 1, N>::backward(A, b, my_invd, x, xs, lane);
    }
  }
};


// Measured and rejected: the same elimination two pivots at a time (2x2 diagonal blocks inverted in closed form, both columns
// published and fetched together, one rank-2 update per row) to halve the number of dependent publish -> fetch -> reciprocal
// chains.  Same-box A/B at W = 10: solve + residual-sweep launch 34.8 us instead of 28.4 us -- the block step's own chain (two
// products for the multipliers, the determinant, twice the LDS reads in the queue ahead of the next fetch) is longer than two
// scalar steps with look-ahead.  Also rejected: no LDS at all -- by symmetry the pivot column is lane K's own row, so every entry can
// be broadcast with v_readlane (static lane and register) and used as the scalar operand of the update; no publish -> fetch round trip,
// but 2 x 27 readlanes per step: 35.4 us in the same A/B.
constexpr int SOLVE_LDS = 128;   // doubles of LDS the solve needs (two pivot-column buffers)

// Rows 6 .. N-1 of a symmetric positive definite system, row i in lane i's registers (A, right-hand side b; damping already
// applied; rows / columns 0..5 are the gauge and are neither read nor written): returns the solution component of this lane
// (0 for lanes < 6 and lanes >= N).  colbuf: SOLVE_LDS doubles of LDS.  One wave.
template <int N>
__device__ __forceinline__ double dense_solve_rows(double (&A)[N > 6 ? N : 7], double b, double* colbuf, int lane) {
  const bool row_ok = lane < N;
  double x = 0.0;
  double my_invd = 1.0;
  if constexpr (N > 6) {
    colbuf[lane] = A[6];
    const double d6 = readlane_f64(A[6], 6);
    const double bk6 = readlane_f64(b, 6);
    __builtin_amdgcn_wave_barrier();
    double pre[LM_PRE];
#pragma unroll
    for (int j = 0; j < LM_PRE; j++) pre[j] = (7 + j < N) ? colbuf[7 + j] : 0.0;
    const double invd6 = pivot_rcp_f64(d6);
    LmElim<6, N>::forward(A, b, my_invd, colbuf, lane, row_ok, invd6, bk6, pre);
  }
  LmElim<N - 1, N>::backward(A, b, my_invd, x, nullptr, lane);
  return x;
}

}  // namespace vxk
