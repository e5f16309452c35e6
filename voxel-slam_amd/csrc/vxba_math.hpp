// Per-voxel / per-(voxel,frame) arithmetic of the BA hot path, shared by the HIP kernels.
// Everything here is branch-light straight-line fp64 written for one GPU lane; the functions are
// also compilable by a host C++ compiler (VX_HD expands to nothing) so tests can check the device
// arithmetic term by term on the CPU -- the shipped path is the HIP kernels only.
//
// Mathematics: SURVEY.md Appendix A (restating VoxelSLAM/src/voxel_map.hpp:132-279 and
// tools.hpp:326-363).  Conventions: R row-major r[3*row+col]; symmetric 3x3 as
// [xx xy xz yy yz zz]; cluster = (P sym6, v3, n).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define VX_HD __host__ __device__ __forceinline__
#else
#define VX_HD inline
#endif

namespace vxm {

// ---------------------------------------------------------------------------------------------
// K2: world transform of one body-frame cluster under pose (R, p), accumulated into S.
//   v' = R v + n p ;  P' = R P R^T + (R v) p^T + p (R v)^T + n p p^T        (tools.hpp:357-363)
// ---------------------------------------------------------------------------------------------
VX_HD void transform_accumulate(const double P[6], const double v[3], double n, const double R[9], const double p[3],
                                double SP[6], double Sv[3], double& SN) {
  // RP = R * P (3x3, P symmetric)
  const double Pm[9] = {P[0], P[1], P[2], P[1], P[3], P[4], P[2], P[4], P[5]};
  double RP[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) RP[3 * i + j] = R[3 * i] * Pm[j] + R[3 * i + 1] * Pm[3 + j] + R[3 * i + 2] * Pm[6 + j];
  double Rv[3];
#pragma unroll
  for (int i = 0; i < 3; i++) Rv[i] = R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2];
  const double np[3] = {n * p[0], n * p[1], n * p[2]};
  // symmetric entries (i <= j) of  RP R^T + Rv p^T + p Rv^T + n p p^T
  constexpr int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int i = I[k], j = J[k];
    double rprt = RP[3 * i] * R[3 * j] + RP[3 * i + 1] * R[3 * j + 1] + RP[3 * i + 2] * R[3 * j + 2];
    SP[k] += ((rprt + Rv[i] * p[j]) + p[i] * Rv[j]) + np[i] * p[j];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) Sv[i] += Rv[i] + np[i];
  SN += n;
}

// ---------------------------------------------------------------------------------------------
// The f32 cluster record of BASELINE configs[2] ("clusters also emitted as f32"), RE-CENTRED: [C sym6 | c | n] with c = v / n the
// cluster's own mean and C = P - v v^T / n its second moments about that mean.  Raw moments cannot be rounded to f32 -- a body-frame
// cluster 30 m from the sensor has P ~ n 900 m^2 and a plane thickness of n 4e-4 m^2 hidden in it, below f32's 6e-8 relative step --
// whereas C is a few m^2 at most (round-off ~1e-7 m^2 against a smallest eigenvalue of ~1e-2) and c rounds like a point does (~2 um).
//   world frame:  w = R c + p ;  P' = R C R^T + n w w^T ;  v' = n w          (tools.hpp:357-363 with P = C + n c c^T, v = n c)
// An unobserved (voxel, frame) slot is the all-zero record and contributes exactly nothing: no test needed.
// ---------------------------------------------------------------------------------------------
VX_HD void cluster_to_centred_f32(const double cl[10], float rec[10]) {
  const double n = cl[9];
  if (n == 0.0) {
#pragma unroll
    for (int k = 0; k < 10; k++) rec[k] = 0.0f;
    return;
  }
  const double inv = 1.0 / n;
  const double c[3] = {cl[6] * inv, cl[7] * inv, cl[8] * inv};
  constexpr int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 6; k++) rec[k] = (float)(cl[k] - cl[6 + I[k]] * c[J[k]]);
#pragma unroll
  for (int k = 0; k < 3; k++) rec[6 + k] = (float)c[k];
  rec[9] = (float)n;     // point counts are far below 2^24
}
VX_HD void transform_accumulate_centred(const float rec[10], const double R[9], const double p[3], double SP[6], double Sv[3], double& SN) {
  const double n = (double)rec[9];
  const double c[3] = {(double)rec[6], (double)rec[7], (double)rec[8]};
  const double Cm[9] = {(double)rec[0], (double)rec[1], (double)rec[2], (double)rec[1], (double)rec[3], (double)rec[4], (double)rec[2], (double)rec[4], (double)rec[5]};
  double RC[9], w[3], nw[3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) RC[3 * i + j] = R[3 * i] * Cm[j] + R[3 * i + 1] * Cm[3 + j] + R[3 * i + 2] * Cm[6 + j];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    w[i] = (R[3 * i] * c[0] + R[3 * i + 1] * c[1] + R[3 * i + 2] * c[2]) + p[i];
    nw[i] = n * w[i];
  }
  constexpr int I[6] = {0, 0, 0, 1, 1, 2}, J[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
  for (int k = 0; k < 6; k++) {
    const int i = I[k], j = J[k];
    SP[k] += (RC[3 * i] * R[3 * j] + RC[3 * i + 1] * R[3 * j + 1] + RC[3 * i + 2] * R[3 * j + 2]) + nw[i] * w[j];
  }
#pragma unroll
  for (int i = 0; i < 3; i++) Sv[i] += nw[i];
  SN += n;
}

// ---------------------------------------------------------------------------------------------
// Symmetric 3x3 eigen-decomposition by cyclic Jacobi rotations (fp64, relative accuracy), ascending
// eigenvalues, unit eigenvectors in the columns of U (row-major U[3*row+col]).  Replaces Eigen's
// SelfAdjointEigenSolver<Matrix3d> at voxel_map.hpp:267,1161,1242; eigenvalues agree to round-off, the
// eigenvector SIGN is implementation-defined in both (every use on the path is quadratic in u).
// ---------------------------------------------------------------------------------------------
// 1/x and 1/sqrt(x) to fp64 round-off without the IEEE division / sqrt expansions (3x shorter dependent chains on
// the GPU: hardware estimate + Newton steps); plain divisions on the host.
// Host builds with -DVXM_EMULATE_HW_ESTIMATES (tests/test_device_math_on_host.py) replace v_rcp_f64 / v_rsq_f64 by the exact value
// truncated to 22 mantissa bits -- the instructions' documented accuracy -- so that the Newton refinements below are exercised on the
// CPU as they run on the GPU.
#if !defined(__HIP_DEVICE_COMPILE__) && defined(VXM_EMULATE_HW_ESTIMATES)
#include <cstring>
#define VXM_EST 1
inline double vxm_truncate22(double v) {
  unsigned long long b;
  std::memcpy(&b, &v, 8);
  b &= ~((1ull << 30) - 1ull);
  std::memcpy(&v, &b, 8);
  return v;
}
inline double vxm_est_rcp(double x) { return vxm_truncate22(1.0 / x); }
inline double vxm_est_rsq(double x) { return vxm_truncate22(1.0 / sqrt(x)); }
#elif defined(__HIP_DEVICE_COMPILE__)
#define VXM_EST 1
__device__ __forceinline__ double vxm_est_rcp(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ double vxm_est_rsq(double x) { return __builtin_amdgcn_rsq(x); }
#else
#define VXM_EST 0
#endif
VX_HD double fast_rcp(double x) {
#if VXM_EST
  double r = vxm_est_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
#else
  return 1.0 / x;
#endif
}
VX_HD double fast_rsqrt(double x) {
#if VXM_EST
  double r = vxm_est_rsq(x);
  // two Newton steps: r <- r (1.5 - 0.5 x r^2)
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
#else
  return 1.0 / sqrt(x);
#endif
}

VX_HD void jacobi_rotate(double& app, double& aqq, double& apq, double& arp, double& arq, double* U, int p, int q, bool late) {
  const double g = 100.0 * fabs(apq);
  if (late && (fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq))) {
    apq = 0.0;
    return;
  }
  if (apq == 0.0) return;
  const double h = aqq - app;
  double t;
  if (fabs(h) + g == fabs(h)) {
    t = apq * fast_rcp(h);
  } else {
    const double theta = 0.5 * h * fast_rcp(apq);
    const double th2 = 1.0 + theta * theta;
    t = fast_rcp(fabs(theta) + th2 * fast_rsqrt(th2));
    if (theta < 0.0) t = -t;
  }
  const double c = fast_rsqrt(1.0 + t * t);
  const double s = t * c;
  const double tau = s * fast_rcp(1.0 + c);
  const double hh = t * apq;
  app -= hh;
  aqq += hh;
  apq = 0.0;
  const double g1 = arp, h1 = arq;
  arp = g1 - s * (h1 + g1 * tau);
  arq = h1 + s * (g1 - h1 * tau);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const double up = U[3 * r + p], uq = U[3 * r + q];
    U[3 * r + p] = up - s * (uq + up * tau);
    U[3 * r + q] = uq + s * (up - uq * tau);
  }
}

// Jacobi iteration on (a00..a22) accumulating the rotations into U (row-major, on entry the starting basis);
// `late_from` = first sweep that may flush negligible off-diagonals.
VX_HD void jacobi_sweeps(double& a00, double& a01, double& a02, double& a11, double& a12, double& a22, double* U, int late_from) {
  for (int sweep = 0; sweep < 12; sweep++) {
    const double sm = fabs(a01) + fabs(a02) + fabs(a12);
    if (sm == 0.0) break;
    const bool late = sweep >= late_from;
    jacobi_rotate(a00, a11, a01, a02, a12, U, 0, 1, late);  // (p,q)=(0,1), other index r=2: a_rp=a02, a_rq=a12
    jacobi_rotate(a00, a22, a02, a01, a12, U, 0, 2, late);  // (0,2), r=1: a_rp=a01, a_rq=a12
    jacobi_rotate(a11, a22, a12, a01, a02, U, 1, 2, late);  // (1,2), r=0: a_rp=a01, a_rq=a02
  }
}

// Branch-free Jacobi rotation for the warm-started solver below: the same rotation as jacobi_rotate (t = sgn(theta) / (|theta| +
// sqrt(1 + theta^2)), c, s, tau), no data-dependent control flow (a wave runs every side of a divergent branch).  apq == 0 gives the
// identity; |theta| is clamped so that theta^2 cannot overflow (beyond the clamp t apq is below 1e-150 |apq|).  The tangent needs
// full precision too: the diagonal update app -= t apq is exact only for the exact root (a tangent from one-Newton-step reciprocals
// was tried -- it perturbs the eigenvalues by 1e-13 |apq|, visible whenever the first rotations are large).
VX_HD void jacobi_rotate_bf(double& app, double& aqq, double& apq, double& arp, double& arq, double* U, int p, int q) {
  const bool nz = apq != 0.0;
  const double h = aqq - app;
  const double theta = 0.5 * h * fast_rcp(nz ? apq : 1.0);
  const double at = fmin(fabs(theta), 1e150);
  const double th2 = fma(at, at, 1.0);
  double t = fast_rcp(at + th2 * fast_rsqrt(th2));
  t = nz ? (theta < 0.0 ? -t : t) : 0.0;
  const double c = fast_rsqrt(fma(t, t, 1.0));
  const double s = t * c;
  const double tau = s * fast_rcp(1.0 + c);
  const double hh = t * apq;
  app -= hh;
  aqq += hh;
  apq = 0.0;
  const double g1 = arp, h1 = arq;
  arp = g1 - s * (h1 + g1 * tau);
  arq = h1 + s * (g1 - h1 * tau);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    const double up = U[3 * r + p], uq = U[3 * r + q];
    U[3 * r + p] = up - s * (uq + up * tau);
    U[3 * r + q] = uq + s * (up - uq * tau);
  }
}
// off-diagonal apq negligible against both of its diagonal entries (jacobi_rotate's flush rule)
VX_HD bool jacobi_negligible(double app, double aqq, double apq) {
  const double g = 100.0 * fabs(apq);
  return apq == 0.0 || ((fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq)));
}

VX_HD void sort_eigen(double a00, double a11, double a22, double* U, double lam[3]) {
  // ascending sort (3-element network) with column swaps
  double l0 = a00, l1 = a11, l2 = a22;
#define VXM_SWAPCOL(x, y, cx, cy)                                   \
  if (y < x) {                                                      \
    double tl = x; x = y; y = tl;                                   \
    for (int r = 0; r < 3; r++) { double tu = U[3 * r + cx]; U[3 * r + cx] = U[3 * r + cy]; U[3 * r + cy] = tu; } \
  }
  VXM_SWAPCOL(l0, l1, 0, 1)
  VXM_SWAPCOL(l1, l2, 1, 2)
  VXM_SWAPCOL(l0, l1, 0, 1)
#undef VXM_SWAPCOL
  lam[0] = l0; lam[1] = l1; lam[2] = l2;
}

VX_HD void eig_sym3(const double Cin[6], double lam[3], double U[9]) {
  double a00 = Cin[0], a01 = Cin[1], a02 = Cin[2], a11 = Cin[3], a12 = Cin[4], a22 = Cin[5];
  U[0] = 1; U[1] = 0; U[2] = 0; U[3] = 0; U[4] = 1; U[5] = 0; U[6] = 0; U[7] = 0; U[8] = 1;
  jacobi_sweeps(a00, a01, a02, a11, a12, a22, U, 3);
  sort_eigen(a00, a11, a22, U, lam);
}

// Warm start: between LM iterations the plane of a voxel barely moves, so C is almost diagonal in the basis of the
// PREVIOUS eigenvectors Up (row-major): rotate C' = Up^T C Up (exact similarity), iterate on C' from the identity
// (1-2 sweeps instead of 5-6), and return U = Up V.  Falls back to the cold start if Up is not an orthonormal basis
// (e.g. a cache that was never written).  Same eigen-decomposition to round-off.
VX_HD void eig_sym3_warm(const double Cin[6], const double Up[9], double lam[3], double U[9]) {
  const double n0 = Up[0] * Up[0] + Up[3] * Up[3] + Up[6] * Up[6], n1 = Up[1] * Up[1] + Up[4] * Up[4] + Up[7] * Up[7];
  const double n2 = Up[2] * Up[2] + Up[5] * Up[5] + Up[8] * Up[8];
  const double d01 = Up[0] * Up[1] + Up[3] * Up[4] + Up[6] * Up[7], d02 = Up[0] * Up[2] + Up[3] * Up[5] + Up[6] * Up[8];
  const double d12 = Up[1] * Up[2] + Up[4] * Up[5] + Up[7] * Up[8];
  const bool ortho = fabs(n0 - 1.0) < 1e-6 && fabs(n1 - 1.0) < 1e-6 && fabs(n2 - 1.0) < 1e-6 && fabs(d01) < 1e-6 && fabs(d02) < 1e-6 &&
                     fabs(d12) < 1e-6;
  if (!ortho) { eig_sym3(Cin, lam, U); return; }
  const double Cm[9] = {Cin[0], Cin[1], Cin[2], Cin[1], Cin[3], Cin[4], Cin[2], Cin[4], Cin[5]};
  double M[9];  // M = C Up
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) M[3 * i + j] = Cm[3 * i] * Up[j] + Cm[3 * i + 1] * Up[3 + j] + Cm[3 * i + 2] * Up[6 + j];
  // C' = Up^T M, symmetric entries
  double a00 = Up[0] * M[0] + Up[3] * M[3] + Up[6] * M[6];
  double a01 = Up[0] * M[1] + Up[3] * M[4] + Up[6] * M[7];
  double a02 = Up[0] * M[2] + Up[3] * M[5] + Up[6] * M[8];
  double a11 = Up[1] * M[1] + Up[4] * M[4] + Up[7] * M[7];
  double a12 = Up[1] * M[2] + Up[4] * M[5] + Up[7] * M[8];
  double a22 = Up[2] * M[2] + Up[5] * M[5] + Up[8] * M[8];
  double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  // Three fixed, branch-free sweeps: C' starts almost diagonal (off-diagonals ~ the pose update), cyclic Jacobi then converges
  // quadratically or better -- 1e-3 -> 1e-6 -> 1e-12 -> 1e-24 relative -- so three sweeps leave nothing above round-off, at a cost that
  // does not depend on the slowest lane of the wave (the generic loop below ran until EVERY lane's off-diagonals were exactly zero,
  // every lane executing every branch: 8.7k cycles per wave in the residual sweep, on the critical path behind the solve).  Whoever is
  // not converged after that (degenerate eigenvalues swapping, a cache from far away) continues in the generic loop.
#pragma unroll 1
  for (int sweep = 0; sweep < 3; sweep++) {
    jacobi_rotate_bf(a00, a11, a01, a02, a12, V, 0, 1);
    jacobi_rotate_bf(a00, a22, a02, a01, a12, V, 0, 2);
    jacobi_rotate_bf(a11, a22, a12, a01, a02, V, 1, 2);
  }
  if (!(jacobi_negligible(a00, a11, a01) && jacobi_negligible(a00, a22, a02) && jacobi_negligible(a11, a22, a12)))
    jacobi_sweeps(a00, a01, a02, a11, a12, a22, V, 0);
  double UV[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) UV[3 * i + j] = Up[3 * i] * V[j] + Up[3 * i + 1] * V[3 + j] + Up[3 * i + 2] * V[6 + j];
  sort_eigen(a00, a11, a22, UV, lam);
#pragma unroll
  for (int k = 0; k < 9; k++) U[k] = UV[k];
}

// Covariance of a merged cluster exactly as the reference forms it (voxel_map.hpp:264-267,
// tools.hpp:333-337):  C = P / N - vbar vbar^T,  vbar = v / N.
VX_HD void cluster_cov(const double P[6], const double v[3], double N, double C[6]) {
  const double vb[3] = {v[0] / N, v[1] / N, v[2] / N};
  C[0] = P[0] / N - vb[0] * vb[0];
  C[1] = P[1] / N - vb[0] * vb[1];
  C[2] = P[2] / N - vb[0] * vb[2];
  C[3] = P[3] / N - vb[1] * vb[1];
  C[4] = P[4] / N - vb[1] * vb[2];
  C[5] = P[5] / N - vb[2] * vb[2];
}

// s_k = sqrt(2 / (lambda_k - lambda_0)), k = 1,2: the scale that turns A^T M A (M = sum 2/(l0-lk) u_k u_k^T,
// voxel_map.hpp:172-174) into -G^T G.
VX_HD void gap_scales(const double lam[3], double& s1, double& s2) {
  s1 = sqrt(2.0 / (lam[1] - lam[0]));
  s2 = sqrt(2.0 / (lam[2] - lam[0]));
}

VX_HD void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
VX_HD double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// ---------------------------------------------------------------------------------------------
// K3 per-(voxel a, frame i) block.  With the cached u = u_0, u_1, u_2, s_1, s_2, vbar, 1/N, coe:
//   rows[0][0..5] = sqrt(coe) s_1 u_1^T A_i          (G row 0, columns 6i..6i+5)
//   rows[1][0..5] = sqrt(coe) s_2 u_2^T A_i          (G row 1)
//   rows[2][0..5] = sqrt(coe) (sqrt2 / N) [w_i ; n_i u]   (z row)
//   acc[0..5]    += coe * g_i,  g_i = A_i^T u         (JacT block,  voxel_map.hpp:202-203)
//   acc[6..11]   += coe * Drr (sym6), acc[12..20] += coe * Drt (3x3 row-major), acc[21..26] += coe * Dtt (sym6)
// so that  H = -(sum_a rows^T rows) + blockdiag_i(D_i)   reproduces voxel_map.hpp:176-232.
// A_i = (1/N)[ (R P + t v^T) hat(r) - R c1 | c2 u^T + (c2.u) I ],  r = R^T u, t = p - vbar,
// c1 = hat(P r) + hat(v)(u.t), c2 = R v + n t, w = v x r.
// Row vectors are evaluated without forming A_i.  With  z = P r + (u.t) v :
//   y^T A_L N = m x r - q x z        (q = R^T y, m = P q + (y.t) v)      -- P r and v only ever enter through z,
//   y^T A_R N = (y.c2) u^T + (c2.u) y^T,
//   g_rot N/2 = z x r                (= P r x r + (u.t) w).
// Block-diagonal correction (symmetric):
//   Drr N/2 = sym(hat(z) hat(r)) - hat(r) P hat(r),    sym(hat(a) hat(b)) = (b a^T + a b^T)/2 - (a.b) I   (linear in a),
//   and for symmetric P   -hat(r) P hat(r) = (r.r)(tr P I - P) - tr P r r^T + P r r^T + r (P r)^T - (r.P r) I,
//   so with h = z + 2 P r - tr P r and c0 = (r.r) tr P - z.r - r.P r:
//   Drr N/2 = (r h^T + h r^T)/2 - (r.r) P + c0 I        -- 31 operations instead of the 87 of forming hat(r) P hat(r);
//   Drt = (2/N) w u^T ;  Dtt = (2 n / N) u u^T.
// (Round 1 formed P hat(r) and hat(r) (P hat(r)) explicitly; the closed form is the same matrix -- checked against the
// oracle's acc_evaluate2 in tests/test_device_math_on_host.py -- at 73 % of the fp64 operations, which is what bounds K3.)
// ---------------------------------------------------------------------------------------------
struct VoxelCache {
  double u0[3], u1[3], u2[3];
  double s1, s2;     // gap scales
  double vbar[3];
  double invN;
  double coe;
  double sc;         // sqrt(coe)
};

// one G row: y in {u1, u2}, scale sk
VX_HD void k3_g_row(const double P[6], const double v[3], const double R[9], const double t[3], const double r[3], const double z[3], const double c2[3],
                    double c2u, const double u[3], const double y[3], double sk, double row[6]) {
  double q[3];
#pragma unroll
  for (int j = 0; j < 3; j++) q[j] = R[j] * y[0] + R[3 + j] * y[1] + R[6 + j] * y[2];
  const double yt = dot3(y, t);
  const double m[3] = {P[0] * q[0] + P[1] * q[1] + P[2] * q[2] + yt * v[0], P[1] * q[0] + P[3] * q[1] + P[4] * q[2] + yt * v[1],
                       P[2] * q[0] + P[4] * q[1] + P[5] * q[2] + yt * v[2]};
  // m x r - q x z
  row[0] = sk * ((m[1] * r[2] - m[2] * r[1]) - (q[1] * z[2] - q[2] * z[1]));
  row[1] = sk * ((m[2] * r[0] - m[0] * r[2]) - (q[2] * z[0] - q[0] * z[2]));
  row[2] = sk * ((m[0] * r[1] - m[1] * r[0]) - (q[0] * z[1] - q[1] * z[0]));
  const double yc2 = sk * dot3(y, c2), sc2u = sk * c2u;
#pragma unroll
  for (int j = 0; j < 3; j++) row[3 + j] = yc2 * u[j] + sc2u * y[j];
}

// RT = false: the caller obtains Drt and Dtt elsewhere (K3's narrow-window kernel reads them off spare columns of its
// MFMA tile: with sqrt2 sqrt(coe) u in three padding columns of the z row, S[6i+j][pad+k] = sum_a (2 coe/N) w_j u_k = Drt
// and S[6i+3+j][pad+k] = sum_a (2 coe n/N) u_j u_k = Dtt) and acc[12..26] stay untouched.
// k3_entry hands its three rows to `emit(r, row)` AS EACH IS FINISHED -- z row first (it needs w only), then the two G rows, then the
// block-diagonal terms -- so that a caller that stores them (the narrow-window Hessian sweep: three 16-byte LDS stores per row) has the
// first stores under way while the rest is still being computed.  Same operations in the same order per value as the array form below.
template <bool RT = true, class Emit>
VX_HD void k3_entry_emit(const double P[6], const double v[3], double n, const double R[9], const double p[3], const VoxelCache& vc, double acc[27], Emit&& emit);
struct K3RowsCollector {
  double (*rows)[6];
  VX_HD void operator()(int r, const double row[6]) const { for (int j = 0; j < 6; j++) rows[r][j] = row[j]; }
};
template <bool RT = true>
VX_HD void k3_entry(const double P[6], const double v[3], double n, const double R[9], const double p[3],
                    const VoxelCache& vc, double rows[3][6], double acc[27]) {
  k3_entry_emit<RT>(P, v, n, R, p, vc, acc, K3RowsCollector{rows});
}
template <bool RT, class Emit>
VX_HD void k3_entry_emit(const double P[6], const double v[3], double n, const double R[9], const double p[3],
                         const VoxelCache& vc, double acc[27], Emit&& emit) {
  const double* u = vc.u0;
  const double invN = vc.invN;
  const double sc = vc.sc;
  // r = R^T u, t = p - vbar
  double r[3], t[3];
#pragma unroll
  for (int j = 0; j < 3; j++) r[j] = R[j] * u[0] + R[3 + j] * u[1] + R[6 + j] * u[2];
#pragma unroll
  for (int j = 0; j < 3; j++) t[j] = p[j] - vc.vbar[j];
  const double ut = dot3(u, t);
  const double Pr[3] = {P[0] * r[0] + P[1] * r[1] + P[2] * r[2], P[1] * r[0] + P[3] * r[1] + P[4] * r[2],
                        P[2] * r[0] + P[4] * r[1] + P[5] * r[2]};
  double z[3];
#pragma unroll
  for (int j = 0; j < 3; j++) z[j] = Pr[j] + ut * v[j];
  double w[3];
  cross3(v, r, w);
  double c2[3];
#pragma unroll
  for (int i = 0; i < 3; i++) c2[i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2]) + n * t[i];
  const double c2u = dot3(c2, u);
  // z row, then the G rows for y = u1 and y = u2
  const double isc = invN * sc;
  {
    const double sz = 1.4142135623730951 * isc;
    const double szn = sz * n;
    double row[6];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      row[j] = sz * w[j];
      row[3 + j] = szn * u[j];
    }
    emit(2, row);
  }
  {
    double row[6];
    k3_g_row(P, v, R, t, r, z, c2, c2u, u, vc.u1, vc.s1 * isc, row);
    emit(0, row);
  }
  {
    double row[6];
    k3_g_row(P, v, R, t, r, z, c2, c2u, u, vc.u2, vc.s2 * isc, row);
    emit(1, row);
  }

  // gradient block g = A^T u = (2/N) [ z x r ; c2u u ]
  const double two_invN = 2.0 * invN;
  const double cg = vc.coe * two_invN;
  acc[0] += cg * (z[1] * r[2] - z[2] * r[1]);
  acc[1] += cg * (z[2] * r[0] - z[0] * r[2]);
  acc[2] += cg * (z[0] * r[1] - z[1] * r[0]);
  const double cgu = cg * c2u;
#pragma unroll
  for (int j = 0; j < 3; j++) acc[3 + j] += cgu * u[j];

  // block-diagonal correction D_i
  const double rr = dot3(r, r), trP = (P[0] + P[3]) + P[5];
  const double c0 = (rr * trP - dot3(z, r)) - dot3(Pr, r);
  double h[3];
#pragma unroll
  for (int j = 0; j < 3; j++) h[j] = (z[j] + 2.0 * Pr[j]) - trP * r[j];
  const double cD = cg;   // coe * 2 / N
  // sym6 order xx xy xz yy yz zz
  acc[6] += cD * ((r[0] * h[0] - rr * P[0]) + c0);
  acc[7] += cD * (0.5 * (r[0] * h[1] + r[1] * h[0]) - rr * P[1]);
  acc[8] += cD * (0.5 * (r[0] * h[2] + r[2] * h[0]) - rr * P[2]);
  acc[9] += cD * ((r[1] * h[1] - rr * P[3]) + c0);
  acc[10] += cD * (0.5 * (r[1] * h[2] + r[2] * h[1]) - rr * P[4]);
  acc[11] += cD * ((r[2] * h[2] - rr * P[5]) + c0);
  if (RT) {
    const double cw[3] = {cD * w[0], cD * w[1], cD * w[2]};
#pragma unroll
    for (int j = 0; j < 3; j++) acc[12 + j] += cw[0] * u[j];
#pragma unroll
    for (int j = 0; j < 3; j++) acc[15 + j] += cw[1] * u[j];
#pragma unroll
    for (int j = 0; j < 3; j++) acc[18 + j] += cw[2] * u[j];
    const double cT = cD * n;
    const double cu[3] = {cT * u[0], cT * u[1], cT * u[2]};
    acc[21] += cu[0] * u[0];
    acc[22] += cu[0] * u[1];
    acc[23] += cu[0] * u[2];
    acc[24] += cu[1] * u[1];
    acc[25] += cu[1] * u[2];
    acc[26] += cu[2] * u[2];
  }
}

}  // namespace vxm
