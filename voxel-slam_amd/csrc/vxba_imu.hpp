// Host-side inertial half of the LiDAR-inertial local BA (O(W) work on 15x15 / 30x30 blocks -- it runs on the host
// while the GPU is busy with the voxel sweeps, exactly where the reference runs it on the caller thread next to its
// LiDAR worker threads, voxel_map.hpp:487-499):
//   imu_init / imu_add      IMU_PRE constructor + add_imu                 preintegration.hpp:32-48, 75-135
//   imu_evaluate            IMU_PRE::give_evaluate                        preintegration.hpp:137-212
//   imu_update_state        IMU_PRE::update_state                         preintegration.hpp:296-303
//   li_add_imu_blocks       the IMU loop + imu_coef scaling of LI_BA_Optimizer::divide_thread   voxel_map.hpp:487-507
//   li_hess_plus            LI_BA_Optimizer::hess_plus                    voxel_map.hpp:455-463
//
// Flat formats (include/vxba.h): all matrices column-major.
//   state (VXBA_STATE_LEN = 24): [R 9 | p 3 | v 3 | bg 3 | ba 3 | g 3]
//   imu   (VXBA_IMU_LEN  = 304): [R_delta 9 | p_delta 3 | v_delta 3 | bg 3 | ba 3 | R_bg 9 | p_bg 9 | p_ba 9 | v_bg 9 |
//                                 v_ba 9 | dtime | dbg 3 | dba 3 | dbg_buf 3 | dba_buf 3 | cov 15x15]
#pragma once
#include <cmath>
#include <cstring>

// the small fixed-size pieces compile for the device as well (the device-resident LiDAR-inertial loop of rounds 1-3 used them)
#if defined(__HIPCC__)
#define VXI_FN __host__ __device__ inline
#else
#define VXI_FN inline
#endif

namespace vxi {

constexpr int DIM = 15, DVEL = 6, STATE_LEN = 24, IMU_LEN = 304;
enum ImuOff { O_RD = 0, O_PD = 9, O_VD = 12, O_BG = 15, O_BA = 18, O_RBG = 21, O_PBG = 30, O_PBA = 39, O_VBG = 48, O_VBA = 57,
              O_DT = 66, O_DBG = 67, O_DBA = 70, O_DBGB = 73, O_DBAB = 76, O_COV = 79 };
enum StOff { S_R = 0, S_P = 9, S_V = 12, S_BG = 15, S_BA = 18, S_G = 21 };

// ---- 3x3 column-major kernels: element (r, c) at [3 c + r] ----
VXI_FN void m3_mul(const double* A, const double* B, double* C) {   // C = A B  (C may not alias)
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) C[3 * c + r] = A[r] * B[3 * c] + A[3 + r] * B[3 * c + 1] + A[6 + r] * B[3 * c + 2];
}
VXI_FN void m3_tmul(const double* A, const double* B, double* C) {  // C = A^T B
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) C[3 * c + r] = A[3 * r] * B[3 * c] + A[3 * r + 1] * B[3 * c + 1] + A[3 * r + 2] * B[3 * c + 2];
}
VXI_FN void m3_vec(const double* A, const double* x, double* y) {   // y = A x
  for (int r = 0; r < 3; r++) y[r] = A[r] * x[0] + A[3 + r] * x[1] + A[6 + r] * x[2];
}
VXI_FN void m3_tvec(const double* A, const double* x, double* y) {  // y = A^T x
  for (int r = 0; r < 3; r++) y[r] = A[3 * r] * x[0] + A[3 * r + 1] * x[1] + A[3 * r + 2] * x[2];
}
VXI_FN void m3_t(const double* A, double* T) {
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) T[3 * c + r] = A[3 * r + c];
}
VXI_FN void m3_hat(const double* v, double* H) {
  H[0] = 0; H[3] = -v[2]; H[6] = v[1];
  H[1] = v[2]; H[4] = 0; H[7] = -v[0];
  H[2] = -v[1]; H[5] = v[0]; H[8] = 0;
}
VXI_FN void m3_eye(double* I) { std::memset(I, 0, 9 * sizeof(double)); I[0] = I[4] = I[8] = 1.0; }

// I + sin(a) K + (1 - cos a) K^2 for unit axis k, angle a
VXI_FN void rodrigues_axis_angle(const double* k, double a, double* R) {
  double K[9], K2[9];
  m3_hat(k, K);
  m3_mul(K, K, K2);
  const double s = std::sin(a), c1 = 1.0 - std::cos(a);
  m3_eye(R);
  for (int q = 0; q < 9; q++) R[q] += s * K[q] + c1 * K2[q];
}
// tools.hpp:51-66
VXI_FN void so3_exp(const double* w, double* R) {
  const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (n >= 1e-11) { const double k[3] = {w[0] / n, w[1] / n, w[2] / n}; rodrigues_axis_angle(k, n, R); }
  else m3_eye(R);
}
// tools.hpp:68-84 (angular velocity * dt, cut-off on the velocity norm)
VXI_FN void so3_exp_dt(const double* w, double dt, double* R) {
  const double n = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
  if (n > 1e-7) { const double k[3] = {w[0] / n, w[1] / n, w[2] / n}; rodrigues_axis_angle(k, n * dt, R); }
  else m3_eye(R);
}
// tools.hpp:86-91
VXI_FN void so3_log(const double* R, double* w) {
  const double tr = R[0] + R[4] + R[8];
  const double theta = (tr > 3.0 - 1e-6) ? 0.0 : std::acos(0.5 * (tr - 1));
  const double K[3] = {R[3 * 1 + 2] - R[3 * 2 + 1], R[3 * 2 + 0] - R[3 * 0 + 2], R[3 * 0 + 1] - R[3 * 1 + 0]};
  const double f = (std::fabs(theta) < 0.001) ? 0.5 : 0.5 * theta / std::sin(theta);
  for (int k = 0; k < 3; k++) w[k] = f * K[k];
}
// right Jacobian of SO(3), tools.hpp:102-116
VXI_FN void so3_jr(const double* v, double* J) {
  const double a = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (a < 1e-9) { m3_eye(J); return; }
  const double k[3] = {v[0] / a, v[1] / a, v[2] / a};
  const double ra = std::sin(a) / a, hc = (1 - std::cos(a)) / a;
  double H[9];
  m3_hat(k, H);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) J[3 * c + r] = (r == c ? ra : 0.0) + (1 - ra) * k[r] * k[c] - hc * H[3 * c + r];
}
// inverse right Jacobian from the rotation matrix, tools.hpp:118-133.  Angle / axis are taken through a unit
// quaternion (angle = 2 atan2(|q_v|, |q_w|) in [0, pi], axis = q_v / |q_v| with the sign of q_w), the convention
// of the reference's Eigen::AngleAxisd(Matrix3d).
VXI_FN void so3_jr_inv(const double* R, double* J) {
  auto E = [&](int r, int c) { return R[3 * c + r]; };
  double q[4];  // x y z w
  const double t = E(0, 0) + E(1, 1) + E(2, 2);
  if (t > 0.0) {
    const double s = std::sqrt(t + 1.0), h = 0.5 / s;
    q[3] = 0.5 * s;
    q[0] = (E(2, 1) - E(1, 2)) * h; q[1] = (E(0, 2) - E(2, 0)) * h; q[2] = (E(1, 0) - E(0, 1)) * h;
  } else {
    int i = 0;
    if (E(1, 1) > E(0, 0)) i = 1;
    if (E(2, 2) > E(i, i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    const double s = std::sqrt(E(i, i) - E(j, j) - E(k, k) + 1.0), h = 0.5 / s;
    q[i] = 0.5 * s;
    q[3] = (E(k, j) - E(j, k)) * h;
    q[j] = (E(j, i) + E(i, j)) * h;
    q[k] = (E(k, i) + E(i, k)) * h;
  }
  double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  if (n == 0.0) { m3_eye(J); return; }
  const double ang = 2.0 * std::atan2(n, std::fabs(q[3]));
  if (q[3] < 0) n = -n;
  if (ang < 1e-9) { m3_eye(J); return; }
  const double a[3] = {q[0] / n, q[1] / n, q[2] / n};
  const double ctt = ang / 2 / std::tan(ang / 2);
  double H[9];
  m3_hat(a, H);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) J[3 * c + r] = (r == c ? ctt : 0.0) + (1 - ctt) * a[r] * a[c] + 0.5 * ang * H[3 * c + r];
}

// ---- small dense column-major helpers (leading dimension = rows) ----
// C (m x n) = A (m x k) * B (k x n)
inline void dm_mul(int m, int k, int n, const double* A, const double* B, double* C) {
  for (int j = 0; j < n; j++) {
    double* cj = C + (size_t)j * m;
    for (int i = 0; i < m; i++) cj[i] = 0.0;
    for (int p = 0; p < k; p++) {
      const double b = B[(size_t)j * k + p];
      if (b == 0.0) continue;
      const double* ap = A + (size_t)p * m;
      for (int i = 0; i < m; i++) cj[i] += ap[i] * b;
    }
  }
}
// C (k x n) = A^T (A is m x k) * B (m x n)
inline void dm_tmul(int m, int k, int n, const double* A, const double* B, double* C) {
  for (int j = 0; j < n; j++)
    for (int i = 0; i < k; i++) {
      double s = 0.0;
      const double* ai = A + (size_t)i * m;
      const double* bj = B + (size_t)j * m;
      for (int p = 0; p < m; p++) s += ai[p] * bj[p];
      C[(size_t)j * k + i] = s;
    }
}
// n x n inverse by LU with row pivoting (the reference calls Eigen's Matrix<15,15>::inverse(), preintegration.hpp:166).
// Returns false on an exactly singular pivot.
inline bool dm_inverse(int n, const double* A, double* inv, double* lu /* n*n */, int* perm /* n */) {
  std::memcpy(lu, A, sizeof(double) * n * n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int piv = k;
    double big = std::fabs(lu[(size_t)k * n + k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(lu[(size_t)k * n + i]) > big) { big = std::fabs(lu[(size_t)k * n + i]); piv = i; }
    if (big == 0.0) return false;
    if (piv != k) {
      for (int j = 0; j < n; j++) { const double t = lu[(size_t)j * n + k]; lu[(size_t)j * n + k] = lu[(size_t)j * n + piv]; lu[(size_t)j * n + piv] = t; }
      const int t = perm[k]; perm[k] = perm[piv]; perm[piv] = t;
    }
    const double d = lu[(size_t)k * n + k];
    for (int i = k + 1; i < n; i++) lu[(size_t)k * n + i] /= d;
    for (int j = k + 1; j < n; j++) {
      const double ukj = lu[(size_t)j * n + k];
      if (ukj == 0.0) continue;
      for (int i = k + 1; i < n; i++) lu[(size_t)j * n + i] -= lu[(size_t)k * n + i] * ukj;
    }
  }
  for (int c = 0; c < n; c++) {
    double* x = inv + (size_t)c * n;
    for (int i = 0; i < n; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;
    for (int j = 0; j < n; j++) {          // forward: unit lower
      const double xj = x[j];
      if (xj == 0.0) continue;
      for (int i = j + 1; i < n; i++) x[i] -= lu[(size_t)j * n + i] * xj;
    }
    for (int j = n - 1; j >= 0; j--) {     // backward: upper
      x[j] /= lu[(size_t)j * n + j];
      const double xj = x[j];
      for (int i = 0; i < j; i++) x[i] -= lu[(size_t)j * n + i] * xj;
    }
  }
  return true;
}
VXI_FN void put33(double* M, int ld, int r0, int c0, const double* B, double scale = 1.0) {
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) M[(size_t)(c0 + c) * ld + r0 + r] = scale * B[3 * c + r];
}

// ---- IMU_PRE ----
inline void imu_init(double* f, const double* bg, const double* ba) {
  std::memset(f, 0, sizeof(double) * IMU_LEN);
  m3_eye(f + O_RD);
  for (int k = 0; k < 3; k++) { f[O_BG + k] = bg ? bg[k] : 0.0; f[O_BA + k] = ba ? ba[k] : 0.0; }
}

// One bias-corrected mid-point sample (preintegration.hpp:75-135).  noise_meas / noise_walk: 6x6 column-major.
inline void imu_add(double* f, const double* gyr, const double* acc, double dt, const double* noise_meas, const double* noise_walk) {
  double* Rd = f + O_RD;
  double Rinc[9], RincT[9], Rjr[9], Rdt[9], Rdt2[9], askew[9], T1[9], T2[9];
  f[O_DT] += dt;
  so3_exp_dt(gyr, dt, Rinc);
  m3_t(Rinc, RincT);
  const double gdt[3] = {gyr[0] * dt, gyr[1] * dt, gyr[2] * dt};
  so3_jr(gdt, Rjr);
  for (int q = 0; q < 9; q++) { Rdt[q] = dt * Rd[q]; Rdt2[q] = 0.5 * dt * dt * Rd[q]; }
  m3_hat(acc, askew);

  double* Rbg = f + O_RBG; double* pbg = f + O_PBG; double* pba = f + O_PBA; double* vbg = f + O_VBG; double* vba = f + O_VBA;
  // bias Jacobians, in the reference's order (each line reads the values the previous lines left)
  for (int q = 0; q < 9; q++) pba[q] += vba[q] * dt - Rdt2[q];
  m3_mul(Rdt2, askew, T1); m3_mul(T1, Rbg, T2);
  for (int q = 0; q < 9; q++) pbg[q] += vbg[q] * dt - T2[q];
  for (int q = 0; q < 9; q++) vba[q] -= Rdt[q];
  m3_mul(Rdt, askew, T1); m3_mul(T1, Rbg, T2);
  for (int q = 0; q < 9; q++) vbg[q] -= T2[q];
  m3_mul(RincT, Rbg, T1);
  for (int q = 0; q < 9; q++) Rbg[q] = T1[q] - Rjr[q] * dt;

  // covariance propagation of the (phi, p, v) block and random walk of the bias block
  double A[81], B[54], C9[81], AC[81], ACA[81], BN[54], BNB[81], At[81], Bt[54];
  std::memset(A, 0, sizeof A); std::memset(B, 0, sizeof B);
  for (int i = 0; i < 9; i++) A[9 * i + i] = 1.0;
  put33(A, 9, 0, 0, RincT);
  m3_mul(Rdt2, askew, T1); put33(A, 9, 3, 0, T1, -1.0);
  double I3[9]; m3_eye(I3);
  put33(A, 9, 3, 6, I3, dt);
  m3_mul(Rdt, askew, T1); put33(A, 9, 6, 0, T1, -1.0);
  put33(B, 9, 0, 0, Rjr, dt);
  put33(B, 9, 3, 3, Rdt2);
  put33(B, 9, 6, 3, Rdt);
  double* cov = f + O_COV;
  for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) C9[9 * c + r] = cov[DIM * c + r];
  for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) At[9 * c + r] = A[9 * r + c];
  for (int c = 0; c < 9; c++) for (int r = 0; r < 6; r++) Bt[6 * c + r] = B[9 * r + c];
  dm_mul(9, 9, 9, A, C9, AC); dm_mul(9, 9, 9, AC, At, ACA);
  dm_mul(9, 6, 6, B, noise_meas, BN); dm_mul(9, 6, 9, BN, Bt, BNB);
  for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) cov[DIM * c + r] = ACA[9 * c + r] + BNB[9 * c + r];
  for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) cov[DIM * (9 + c) + 9 + r] += noise_walk[6 * c + r] * dt;

  // the preintegrated measurement itself
  double a2[3], a1[3];
  m3_vec(Rdt2, acc, a2);
  m3_vec(Rdt, acc, a1);
  for (int k = 0; k < 3; k++) { f[O_PD + k] += f[O_VD + k] * dt + a2[k]; }
  for (int k = 0; k < 3; k++) f[O_VD + k] += a1[k];
  m3_mul(Rd, Rinc, T1);
  std::memcpy(Rd, T1, sizeof T1);
}

constexpr int NCG = 2 * DIM + 3;  // Jacobian columns with the gravity block (give_evaluate_g)
struct ImuWork {
  double joc[DIM * NCG];          // 15 x 30 (or 15 x 33 with gravity)
  double cov_inv[DIM * DIM], lu[DIM * DIM];
  double ci_j[DIM * NCG];         // cov^-1 J
  int perm[DIM];
};

// r^T cov^-1 r; with jac: jtj (30x30) = J^T cov^-1 J, gg (30) = J^T cov^-1 r   (preintegration.hpp:137-212).
// with_g: give_evaluate_g (:214-294) -- three more Jacobian columns for the gravity vector, jtj 33x33, gg 33.
// cov_inv_cached: the factor's 15x15 information matrix if the caller already inverted cov (it does not change while an LM
// loop runs; upstream re-inverts it in every evaluation).
// The fixed-size part of give_evaluate[_g]: the 15-dimensional residual rr and, with jac, the 15 x nc Jacobian J (column-major, ld 15;
// nc = 30, or 33 with the gravity columns) -- preintegration.hpp:137-201 / 214-283.  Shared by the host shell and the device loop.
VXI_FN void imu_residual_jac(const double* f, const double* s1, const double* s2, bool jac, bool with_g, double* rr, double* J) {
  const double* R1 = s1 + S_R; const double* R2 = s2 + S_R;
  const double dt = f[O_DT];
  double rb[3], Eb[9], Rc[9], tc[3], vc[3], t3[3], u3[3];
  m3_vec(f + O_RBG, f + O_DBG, rb);
  so3_exp(rb, Eb);
  m3_mul(f + O_RD, Eb, Rc);
  m3_vec(f + O_PBG, f + O_DBG, t3); m3_vec(f + O_PBA, f + O_DBA, u3);
  for (int k = 0; k < 3; k++) tc[k] = f[O_PD + k] + t3[k] + u3[k];
  m3_vec(f + O_VBG, f + O_DBG, t3); m3_vec(f + O_VBA, f + O_DBA, u3);
  for (int k = 0; k < 3; k++) vc[k] = f[O_VD + k] + t3[k] + u3[k];

  double R12[9], res_r[9], dv[3], dp[3], exp_v[3], exp_t[3];
  m3_tmul(R1, R2, R12);
  m3_tmul(Rc, R12, res_r);
  for (int k = 0; k < 3; k++) {
    dv[k] = s2[S_V + k] - s1[S_V + k] - dt * s1[S_G + k];
    dp[k] = s2[S_P + k] - s1[S_P + k] - s1[S_V + k] * dt - 0.5 * dt * dt * s1[S_G + k];
  }
  m3_tvec(R1, dv, exp_v);
  m3_tvec(R1, dp, exp_t);
  so3_log(res_r, rr);
  for (int k = 0; k < 3; k++) {
    rr[3 + k] = exp_t[k] - tc[k];
    rr[6 + k] = exp_v[k] - vc[k];
    rr[9 + k] = s2[S_BG + k] - s1[S_BG + k];
    rr[12 + k] = s2[S_BA + k] - s1[S_BA + k];
  }
  if (jac) {
    // column block a = st1 (cols 0..14), b = st2 (cols 15..29); ld = 15
    for (int q = 0; q < DIM * (with_g ? NCG : 2 * DIM); q++) J[q] = 0.0;
    double Jri[9], T1[9], T2[9], R1t[9], rrT[9], Jrb[9], H[9], I3[9];
    m3_eye(I3);
    so3_jr_inv(res_r, Jri);
    m3_t(R1, R1t);
    m3_tmul(R2, R1, T1); m3_mul(Jri, T1, T2);                       // JR_inv R2^T R1
    put33(J, DIM, 0, 0, T2, -1.0);
    put33(J, DIM, 0, DIM, Jri);
    m3_t(res_r, rrT); so3_jr(rb, Jrb);
    m3_mul(Jri, rrT, T1); m3_mul(T1, Jrb, T2); m3_mul(T2, f + O_RBG, T1);
    put33(J, DIM, 0, 9, T1, -1.0);

    m3_hat(exp_t, H); put33(J, DIM, 3, 0, H);
    put33(J, DIM, 3, 3, R1t, -1.0);
    put33(J, DIM, 3, 6, R1t, -dt);
    put33(J, DIM, 3, 9, f + O_PBG, -1.0);
    put33(J, DIM, 3, 12, f + O_PBA, -1.0);
    put33(J, DIM, 3, DIM + 3, R1t);

    m3_hat(exp_v, H); put33(J, DIM, 6, 0, H);
    put33(J, DIM, 6, 6, R1t, -1.0);
    put33(J, DIM, 6, 9, f + O_VBG, -1.0);
    put33(J, DIM, 6, 12, f + O_VBA, -1.0);
    put33(J, DIM, 6, DIM + 6, R1t);

    put33(J, DIM, 9, 9, I3, -1.0);
    put33(J, DIM, 12, 12, I3, -1.0);
    put33(J, DIM, 9, DIM + 9, I3);
    put33(J, DIM, 12, DIM + 12, I3);
    if (with_g) {   // preintegration.hpp:277-278
      put33(J, DIM, 3, 2 * DIM, R1t, -0.5 * dt * dt);
      put33(J, DIM, 6, 2 * DIM, R1t, -dt);
    }

  }
}

// The products of give_evaluate on top of (rr, J):  q = cov^-1 r,  M = cov^-1 J,  jtj = J^T M,  gg = J^T q,  returns r . q.
// Fixed sizes and unit-stride inner loops (the compiler turns them into SSE2 / AVX2 code; vxi::li_add_imu_blocks picks an AVX2 + FMA
// instance at run time): M column by column as a combination of the columns of cov^-1 that J's non-zeros select (18 of J's 50 3x3
// blocks are set), jtj column by column as a combination of the ROWS of J.  Every sum runs over the same index in the same (ascending)
// order as the plain triple loops; gg associates J^T (cov^-1 r) where upstream writes (J^T cov^-1) r -- round-off only.
template <int NC>
__attribute__((always_inline)) inline double imu_products(const double* J /* 15 x NC, column-major */, const double* ci /* 15 x 15 */, const double* rr, bool jac,
                                                          double* jtj /* NC x NC, column-major */, double* gg) {
  double q[DIM];
  for (int i = 0; i < DIM; i++) q[i] = 0.0;
  for (int k = 0; k < DIM; k++) {
    const double r = rr[k];
    const double* ck = ci + (size_t)k * DIM;
    for (int i = 0; i < DIM; i++) q[i] += ck[i] * r;      // q(i) = sum_k cov_inv(i, k) r(k)
  }
  double res = 0.0;
  for (int i = 0; i < DIM; i++) res += rr[i] * q[i];
  if (!jac) return res;
  double Jr[DIM][NC];                                       // rows of J
  for (int j = 0; j < NC; j++)
    for (int b = 0; b < DIM; b++) Jr[b][j] = J[(size_t)j * DIM + b];
  for (int j = 0; j < NC; j++) {
    double m[DIM];
    for (int i = 0; i < DIM; i++) m[i] = 0.0;
    const double* jc = J + (size_t)j * DIM;
    double g = 0.0;
    for (int p = 0; p < DIM; p++) {
      const double b = jc[p];
      if (b == 0.0) continue;
      const double* cp = ci + (size_t)p * DIM;
      for (int i = 0; i < DIM; i++) m[i] += cp[i] * b;      // M(:, j) = sum_p cov_inv(:, p) J(p, j)
      g += b * q[p];
    }
    gg[j] = g;
    double* out = jtj + (size_t)j * NC;
    for (int i = 0; i < NC; i++) out[i] = 0.0;
    for (int b = 0; b < DIM; b++) {
      const double mb = m[b];
      for (int i = 0; i < NC; i++) out[i] += Jr[b][i] * mb;  // jtj(:, j) = sum_b J(b, :)^T M(b, j)
    }
  }
  return res;
}

__attribute__((always_inline)) inline double imu_evaluate(const double* f, const double* s1, const double* s2, bool jac, double* jtj, double* gg, ImuWork& w, bool* ok = nullptr,
                           bool with_g = false, const double* cov_inv_cached = nullptr) {
  double rr[DIM];
  imu_residual_jac(f, s1, s2, jac, with_g, rr, w.joc);
  bool inv_ok = true;
  const double* ci = cov_inv_cached;
  if (!ci) { inv_ok = dm_inverse(DIM, f + O_COV, w.cov_inv, w.lu, w.perm); ci = w.cov_inv; }
  if (ok) *ok = inv_ok;
  return with_g ? imu_products<NCG>(w.joc, ci, rr, jac, jtj, gg) : imu_products<2 * DIM>(w.joc, ci, rr, jac, jtj, gg);
}

VXI_FN void imu_update_state(double* f, const double* dxi15) {
  for (int k = 0; k < 3; k++) {
    f[O_DBGB + k] = f[O_DBG + k];
    f[O_DBAB + k] = f[O_DBA + k];
    f[O_DBG + k] += dxi15[9 + k];
    f[O_DBA + k] += dxi15[12 + k];
  }
}
VXI_FN void imu_rollback(double* f) {   // voxel_map.hpp:639-643
  for (int k = 0; k < 3; k++) { f[O_DBG + k] = f[O_DBGB + k]; f[O_DBA + k] = f[O_DBAB + k]; }
}

// Hess (n x n, n = 15W [+3 with gravity], zeroed by the caller) += imu blocks, then everything scaled by imu_coef; returns
// the scaled residual.  with_g: the gravity rows / columns sit at the tail (voxel_map.hpp:700-711).
// The scaling pass (upstream: Hess *= imu_coef over the whole matrix, :504) only visits what the factors can have touched: the
// rows of frames j-1 .. j+1 in the columns of frame j, and the gravity rows / columns -- everything else is still the caller's zero.
__attribute__((always_inline)) inline double li_add_imu_blocks_impl(int W, const double* states, const double* imus, double imu_coef, bool jac, double* Hess, double* JacT,
                                                                    ImuWork& w, bool* ok, bool with_g, const double* cov_invs /* (W-1) x 225 */) {
  const int n = DIM * W + (with_g ? 3 : 0), nc = with_g ? NCG : 2 * DIM, gq = DIM * W;
  double jtj[NCG * NCG], gg[NCG];
  double residual = 0.0;
  bool all_ok = true;
  for (int i = 0; i < W - 1; i++) {
    bool one_ok = true;
    residual += imu_evaluate(imus + (size_t)IMU_LEN * i, states + STATE_LEN * i, states + STATE_LEN * (i + 1), jac, jtj, gg, w, &one_ok, with_g,
                             cov_invs ? cov_invs + (size_t)DIM * DIM * i : nullptr);
    all_ok = all_ok && one_ok;
    if (jac) {
      for (int c = 0; c < 2 * DIM; c++) {
        double* hc = Hess + (size_t)(i * DIM + c) * n + i * DIM;
        const double* jc = jtj + (size_t)c * nc;
        for (int r = 0; r < 2 * DIM; r++) hc[r] += jc[r];
      }
      for (int r = 0; r < 2 * DIM; r++) JacT[i * DIM + r] += gg[r];
      if (with_g) {
        for (int c = 0; c < 3; c++)
          for (int r = 0; r < 2 * DIM; r++) {
            Hess[(size_t)(gq + c) * n + i * DIM + r] += jtj[(size_t)(2 * DIM + c) * nc + r];
            Hess[(size_t)(i * DIM + r) * n + gq + c] += jtj[(size_t)r * nc + 2 * DIM + c];
          }
        for (int c = 0; c < 3; c++)
          for (int r = 0; r < 3; r++) Hess[(size_t)(gq + c) * n + gq + r] += jtj[(size_t)(2 * DIM + c) * nc + 2 * DIM + r];
        for (int r = 0; r < 3; r++) JacT[gq + r] += gg[2 * DIM + r];
      }
    }
  }
  if (jac) {
    for (int c = 0; c < gq; c++) {
      const int j = c / DIM;
      const int r0 = (j > 0 ? j - 1 : 0) * DIM, r1 = (j + 2 < W ? j + 2 : W) * DIM;
      double* hc = Hess + (size_t)c * n;
      for (int r = r0; r < r1; r++) hc[r] *= imu_coef;
      for (int r = gq; r < n; r++) hc[r] *= imu_coef;
    }
    for (int c = gq; c < n; c++) {
      double* hc = Hess + (size_t)c * n;
      for (int r = 0; r < n; r++) hc[r] *= imu_coef;
    }
    for (int k = 0; k < n; k++) JacT[k] *= imu_coef;
  }
  if (ok) *ok = all_ok;
  return residual * (imu_coef * 0.5);
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target("avx2,fma"))) inline double li_add_imu_blocks_avx2(int W, const double* states, const double* imus, double imu_coef, bool jac, double* Hess,
                                                                         double* JacT, ImuWork& w, bool* ok, bool with_g, const double* cov_invs) {
  return li_add_imu_blocks_impl(W, states, imus, imu_coef, jac, Hess, JacT, w, ok, with_g, cov_invs);
}
#endif
inline double li_add_imu_blocks(int W, const double* states, const double* imus, double imu_coef, bool jac, double* Hess, double* JacT,
                                ImuWork& w, bool* ok, bool with_g = false, const double* cov_invs = nullptr /* (W-1) x 225 */) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  static const bool avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  if (avx2) return li_add_imu_blocks_avx2(W, states, imus, imu_coef, jac, Hess, JacT, w, ok, with_g, cov_invs);
#endif
  return li_add_imu_blocks_impl(W, states, imus, imu_coef, jac, Hess, JacT, w, ok, with_g, cov_invs);
}

// information matrices of all factors of a window, once per LM loop; false if one covariance is singular
inline bool li_invert_covariances(int W, const double* imus, double* cov_invs) {
  double lu[DIM * DIM];
  int perm[DIM];
  bool ok = true;
  for (int i = 0; i < W - 1; i++) ok = dm_inverse(DIM, imus + (size_t)IMU_LEN * i + O_COV, cov_invs + (size_t)DIM * DIM * i, lu, perm) && ok;
  return ok;
}

// scatter the 6W LiDAR system into the 15W (+3) one (voxel_map.hpp:455-463, 663-671); n = leading dimension of Hess
inline void li_hess_plus(int W, double* Hess, double* JacT, const double* hs, const double* js, int n = 0) {
  const int m = DVEL * W;
  if (n == 0) n = DIM * W;
  for (int i = 0; i < W; i++) {
    for (int k = 0; k < DVEL; k++) JacT[i * DIM + k] += js[i * DVEL + k];
    for (int j = 0; j < W; j++)
      for (int c = 0; c < DVEL; c++)
        for (int r = 0; r < DVEL; r++) Hess[(size_t)(j * DIM + c) * n + i * DIM + r] += hs[(size_t)(j * DVEL + c) * m + i * DVEL + r];
  }
}

}  // namespace vxi
