// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  PINNED against the reference's own code compiled here (oracle/_ref/libref.so, `make -C oracle ref`: tests/test_ref_pin.py):
// IMU_PRE (add_imu, give_evaluate[_g], update_state), jr / jr_inv, LI_BA_Optimizer and its gravity variant through the unmodified
// preintegration.hpp / voxel_map.hpp.  Matrix<15,15>::inverse() inside libref is this file's restatement (Eigen is absent): circular for that one.
//
// CPU restatement of the inertial half of the local BA:
//   * IMU preintegration factor  -- class IMU_PRE, preintegration.hpp:11-310
//       add_imu         :75-135   (mid-point sample already bias-corrected by the caller, push_imu :50-73)
//       give_evaluate   :137-212  (15-dim residual, 15x30 Jacobian, information = cov^-1)
//       update_state    :296-303
//   * SO(3) helpers jr / jr_inv / Exp(w, dt)   tools.hpp:68-133
//   * LI_BA_Optimizer  voxel_map.hpp:446-655  (hess_plus, divide_thread, only_residual, damping_iter)
//
// Third-party arithmetic restated from its published form (Eigen 3.3.7, not vendored):
//   * Matrix<15,15>::inverse()  -> PartialPivLU based inverse (preintegration.hpp:166): LU with row pivoting,
//     then solve against the identity;
//   * AngleAxisd(Matrix3d)      -> via a unit quaternion (Shepperd's branch on the trace), angle = 2 atan2(|vec|, |w|),
//     axis = vec / (+-|vec|) with the sign of w (tools.hpp:118-133).
#pragma once
#include <deque>

#include "vxo_ba.hpp"

namespace vxo {

constexpr int DIM = 15;   // [dphi, dp, dv, dbg, dba] per frame (tools.hpp:154-162)
constexpr int DVEL = 6;   // LiDAR block per frame (voxel_map.hpp:448)

// ---- dense helpers on MatX -------------------------------------------------
inline MatX mat_mul(const MatX& A, const MatX& B) {
  MatX C(A.rows, B.cols);
  for (int j = 0; j < B.cols; j++)
    for (int k = 0; k < A.cols; k++) {
      const double b = B(k, j);
      if (b == 0.0) continue;
      for (int i = 0; i < A.rows; i++) C(i, j) += A(i, k) * b;
    }
  return C;
}
inline MatX mat_t(const MatX& A) {
  MatX T(A.cols, A.rows);
  for (int j = 0; j < A.cols; j++)
    for (int i = 0; i < A.rows; i++) T(j, i) = A(i, j);
  return T;
}
inline void set_block(MatX& A, int r0, int c0, const M3& B) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) A(r0 + r, c0 + c) = B(r, c);
}
// inverse through LU with partial (row) pivoting
inline MatX mat_inverse(const MatX& Ain) {
  const int n = Ain.rows;
  MatX LU = Ain;
  std::vector<int> perm(n);
  for (int i = 0; i < n; i++) perm[i] = i;
  for (int k = 0; k < n; k++) {
    int piv = k;
    double big = std::fabs(LU(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(LU(i, k)) > big) { big = std::fabs(LU(i, k)); piv = i; }
    if (piv != k) {
      for (int j = 0; j < n; j++) std::swap(LU(k, j), LU(piv, j));
      std::swap(perm[k], perm[piv]);
    }
    const double d = LU(k, k);
    for (int i = k + 1; i < n; i++) {
      LU(i, k) /= d;
      const double l = LU(i, k);
      for (int j = k + 1; j < n; j++) LU(i, j) -= l * LU(k, j);
    }
  }
  MatX inv(n, n);
  for (int c = 0; c < n; c++) {
    std::vector<double> x(n);
    for (int i = 0; i < n; i++) x[i] = (perm[i] == c) ? 1.0 : 0.0;            // P e_c
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) x[i] -= LU(i, j) * x[j];          // L^-1
    for (int i = n - 1; i >= 0; i--) { for (int j = i + 1; j < n; j++) x[i] -= LU(i, j) * x[j]; x[i] /= LU(i, i); }   // U^-1
    for (int i = 0; i < n; i++) inv(i, c) = x[i];
  }
  return inv;
}

// ---- SO(3) helpers -----------------------------------------------------------
// tools.hpp:68-84
inline M3 Exp(const V3& ang_vel, double dt) {
  const double n = norm(ang_vel);
  if (n > 1e-7) {
    const V3 axis = ang_vel / n;
    const M3 K = hat(axis);
    const double a = n * dt;
    return eye33() + std::sin(a) * K + ((1.0 - std::cos(a)) * K) * K;
  }
  return eye33();
}
// tools.hpp:102-116
inline M3 jr(V3 vec) {
  const double ang = norm(vec);
  if (ang < 1e-9) return eye33();
  vec = vec / ang;
  const double ra = std::sin(ang) / ang;
  return ra * eye33() + outer((1 - ra) * vec, vec) - ((1 - std::cos(ang)) / ang) * hat(vec);   // ((1-ra)*vec)*vec^T, as the reference's text associates
}
// rotation matrix -> (angle in [0, pi], unit axis), the way Eigen's AngleAxisd(Matrix3d) gets there
inline void angle_axis(const M3& R, double& angle, V3& axis) {
  double w, x, y, z;
  const double t = R(0,0) + R(1,1) + R(2,2);
  if (t > 0.0) {
    double s = std::sqrt(t + 1.0);
    w = 0.5 * s;
    s = 0.5 / s;
    x = (R(2,1) - R(1,2)) * s; y = (R(0,2) - R(2,0)) * s; z = (R(1,0) - R(0,1)) * s;
  } else {
    int i = 0;
    if (R(1,1) > R(0,0)) i = 1;
    if (R(2,2) > R(i,i)) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R(i,i) - R(j,j) - R(k,k) + 1.0);
    double q[3];
    q[i] = 0.5 * s;
    s = 0.5 / s;
    w = (R(k,j) - R(j,k)) * s;
    q[j] = (R(j,i) + R(i,j)) * s;
    q[k] = (R(k,i) + R(i,k)) * s;
    x = q[0]; y = q[1]; z = q[2];
  }
  double n = std::sqrt(x * x + y * y + z * z);
  if (n != 0.0) {
    angle = 2.0 * std::atan2(n, std::fabs(w));
    if (w < 0) n = -n;
    axis = v3(x / n, y / n, z / n);
  } else {
    angle = 0.0;
    axis = v3(1, 0, 0);
  }
}
// tools.hpp:118-133
inline M3 jr_inv(const M3& rotR) {
  double ang; V3 axi;
  angle_axis(rotR, ang, axi);
  if (ang < 1e-9) return eye33();
  const double ctt = ang / 2 / std::tan(ang / 2);
  return ctt * eye33() + outer((1 - ctt) * axi, axi) + (ang / 2) * hat(axi);
}

// tools.hpp:135-199 (the fields the BA touches)
struct ImuState {
  M3 R = eye33();
  V3 p = zero3(), v = zero3(), bg = zero3(), ba = zero3(), g = zero3();
};

// preintegration.hpp:11-310
class IMU_PRE {
 public:
  M3 R_delta = eye33();
  V3 p_delta = zero3(), v_delta = zero3();
  V3 bg = zero3(), ba = zero3();
  M3 R_bg = zero33(), p_bg = zero33(), p_ba = zero33(), v_bg = zero33(), v_ba = zero33();
  double dtime = 0;
  V3 dbg = zero3(), dba = zero3(), dbg_buf = zero3(), dba_buf = zero3();
  MatX cov = MatX(DIM, DIM);

  IMU_PRE() {}
  IMU_PRE(const V3& bg1, const V3& ba1) : bg(bg1), ba(ba1) {}

  // :75-135.  noiseMeas / noiseWalk are the reference's file-scope 6x6 matrices (:9, voxelslam.cpp:828-833).
  void add_imu(const V3& cur_gyr, const V3& cur_acc, double dt, const MatX& noiseMeas, const MatX& noiseWalk) {
    dtime += dt;
    const M3 R_inc = Exp(cur_gyr, dt);
    const M3 R_jr = jr(cur_gyr * dt);
    const M3 R_dt = dt * R_delta;
    const M3 R_dt2_2 = (0.5 * dt * dt) * R_delta;
    const M3 acc_skew = hat(cur_acc);

    p_ba = p_ba + v_ba * dt - R_dt2_2;
    p_bg = p_bg + v_bg * dt - R_dt2_2 * acc_skew * R_bg;
    v_ba = v_ba - R_dt;
    v_bg = v_bg - R_dt * acc_skew * R_bg;
    R_bg = transpose(R_inc) * R_bg - R_jr * dt;

    MatX A(9, 9), B(9, 6);
    for (int i = 0; i < 9; i++) A(i, i) = 1.0;
    set_block(A, 0, 0, transpose(R_inc));
    set_block(A, 3, 0, -1.0 * (R_dt2_2 * acc_skew));
    set_block(A, 3, 6, dt * eye33());
    set_block(A, 6, 0, -1.0 * (R_dt * acc_skew));
    set_block(B, 0, 0, R_jr * dt);
    set_block(B, 3, 3, R_dt2_2);
    set_block(B, 6, 3, R_dt);

    MatX c9(9, 9);
    for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) c9(r, c) = cov(r, c);
    const MatX n9 = mat_mul(mat_mul(A, c9), mat_t(A));
    const MatX b9 = mat_mul(mat_mul(B, noiseMeas), mat_t(B));
    for (int c = 0; c < 9; c++) for (int r = 0; r < 9; r++) cov(r, c) = n9(r, c) + b9(r, c);
    for (int c = 0; c < 6; c++) for (int r = 0; r < 6; r++) cov(9 + r, 9 + c) += noiseWalk(r, c) * dt;

    p_delta = p_delta + v_delta * dt + R_dt2_2 * cur_acc;
    v_delta = v_delta + R_dt * cur_acc;
    R_delta = R_delta * R_inc;
  }

  // :137-212.  jtj (30x30) / gg (30) are written only when jac_enable.
  double give_evaluate(const ImuState& st1, const ImuState& st2, MatX& jtj, std::vector<double>& gg, bool jac_enable) const {
    return evaluate(st1, st2, jtj, gg, jac_enable, false);
  }
  // :214-294.  Same residual; the Jacobian gets three more columns for the gravity vector: jtj 33x33, gg 33.
  double give_evaluate_g(const ImuState& st1, const ImuState& st2, MatX& jtj, std::vector<double>& gg, bool jac_enable) const {
    return evaluate(st1, st2, jtj, gg, jac_enable, true);
  }
  double evaluate(const ImuState& st1, const ImuState& st2, MatX& jtj, std::vector<double>& gg, bool jac_enable, bool with_g) const {
    MatX joca(DIM, DIM), jocb(DIM, DIM);
    std::vector<double> rr(DIM, 0.0);

    const M3 R_correct = R_delta * Exp(R_bg * dbg);
    const V3 t_correct = p_delta + p_bg * dbg + p_ba * dba;
    const V3 v_correct = v_delta + v_bg * dbg + v_ba * dba;

    const M3 R1t = transpose(st1.R);
    const M3 res_r = transpose(R_correct) * R1t * st2.R;
    const V3 exp_v = R1t * (st2.v - st1.v - dtime * st1.g);
    const V3 res_v = exp_v - v_correct;
    const V3 exp_t = R1t * (st2.p - st1.p - st1.v * dtime - (0.5 * dtime * dtime) * st1.g);
    const V3 res_t = exp_t - t_correct;
    const V3 res_bg = st2.bg - st1.bg;
    const V3 res_ba = st2.ba - st1.ba;
    const double b_wei = 1;

    const V3 lr = Log(res_r);
    for (int k = 0; k < 3; k++) {
      rr[k] = lr[k]; rr[3 + k] = res_t[k]; rr[6 + k] = res_v[k]; rr[9 + k] = res_bg[k] * b_wei; rr[12 + k] = res_ba[k] * b_wei;
    }
    const MatX cov_inv = mat_inverse(cov);

    if (jac_enable) {
      const M3 JR_inv = jr_inv(res_r);
      set_block(joca, 0, 0, -1.0 * (JR_inv * transpose(st2.R) * st1.R));
      set_block(jocb, 0, 0, JR_inv);
      set_block(joca, 0, 9, -1.0 * (JR_inv * transpose(res_r) * jr(R_bg * dbg) * R_bg));

      set_block(joca, 3, 0, hat(exp_t));
      set_block(joca, 3, 3, -1.0 * R1t);
      set_block(joca, 3, 6, -1.0 * (R1t * dtime));
      set_block(joca, 3, 9, -1.0 * p_bg);
      set_block(joca, 3, 12, -1.0 * p_ba);
      set_block(jocb, 3, 3, R1t);

      set_block(joca, 6, 0, hat(exp_v));
      set_block(joca, 6, 6, -1.0 * R1t);
      set_block(joca, 6, 9, -1.0 * v_bg);
      set_block(joca, 6, 12, -1.0 * v_ba);
      set_block(jocb, 6, 6, R1t);

      set_block(joca, 9, 9, -b_wei * eye33());
      set_block(joca, 12, 12, -b_wei * eye33());
      set_block(jocb, 9, 9, b_wei * eye33());
      set_block(jocb, 12, 12, b_wei * eye33());

      const int nc = 2 * DIM + (with_g ? 3 : 0);
      MatX joc(DIM, nc);
      for (int c = 0; c < DIM; c++)
        for (int r = 0; r < DIM; r++) { joc(r, c) = joca(r, c); joc(r, DIM + c) = jocb(r, c); }
      if (with_g) {   // :277-278
        set_block(joc, 3, 2 * DIM, R1t * (-0.5 * dtime * dtime));
        set_block(joc, 6, 2 * DIM, R1t * (-dtime));
      }
      const MatX jt_ci = mat_mul(mat_t(joc), cov_inv);   // nc x 15
      jtj = mat_mul(jt_ci, joc);
      gg.assign(nc, 0.0);
      for (int i = 0; i < nc; i++) {
        double s = 0;
        for (int k = 0; k < DIM; k++) s += jt_ci(i, k) * rr[k];
        gg[i] = s;
      }
    }
    double res = 0;
    for (int i = 0; i < DIM; i++) {
      double s = 0;
      for (int k = 0; k < DIM; k++) s += cov_inv(i, k) * rr[k];
      res += rr[i] * s;
    }
    return res;
  }

  // :296-303
  void update_state(const double* dxi15) {
    dbg_buf = dbg;
    dba_buf = dba;
    dbg = dbg + v3(dxi15[9], dxi15[10], dxi15[11]);
    dba = dba + v3(dxi15[12], dxi15[13], dxi15[14]);
  }
};

// voxel_map.hpp:446-655
class LI_BA_Optimizer {
 public:
  int win_size = 0, jac_leng = 0, imu_leng = 0, thd_num = 5;
  double imu_coef = 1e-4;   // voxel_map.hpp:446 / voxelslam.cpp:822
  std::vector<LMTraceEntry> trace;

  // :455-463
  void hess_plus(MatX& Hess, std::vector<double>& JacT, const MatX& hs, const std::vector<double>& js) {
    for (int i = 0; i < win_size; i++) {
      for (int k = 0; k < DVEL; k++) JacT[i * DIM + k] += js[i * DVEL + k];
      for (int j = 0; j < win_size; j++)
        for (int c = 0; c < DVEL; c++)
          for (int r = 0; r < DVEL; r++) Hess(i * DIM + r, j * DIM + c) += hs(i * DVEL + r, j * DVEL + c);
    }
  }

  static std::vector<Pose> poses_of(const std::vector<ImuState>& xs) {
    std::vector<Pose> ps(xs.size());
    for (size_t i = 0; i < xs.size(); i++) { ps[i].R = xs[i].R; ps[i].p = xs[i].p; }
    return ps;
  }

  // :465-523
  double divide_thread(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, MatX& Hess,
                       std::vector<double>& JacT) {
    double residual = 0;
    Hess.setZero();
    std::fill(JacT.begin(), JacT.end(), 0.0);
    std::vector<MatX> hessians(thd_num);
    std::vector<std::vector<double>> jacobins(thd_num);
    std::vector<double> resis(thd_num, 0);
    for (int i = 0; i < thd_num; i++) {
      hessians[i].resize(jac_leng, jac_leng);
      jacobins[i].assign(jac_leng, 0.0);
    }
    int tthd_num = thd_num;
    const int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tthd_num) tthd_num = 1;
    const double part = 1.0 * g_size / tthd_num;
    const std::vector<Pose> poses = poses_of(x_stats);

    std::vector<std::thread*> mthreads(tthd_num, nullptr);
    for (int i = 1; i < tthd_num; i++)
      mthreads[i] = new std::thread(&LidarFactor::acc_evaluate2, &voxhess, poses, (int)(part * i), (int)(part * (i + 1)),
                                    std::ref(hessians[i]), std::ref(jacobins[i]), std::ref(resis[i]));

    MatX jtj(2 * DIM, 2 * DIM);
    std::vector<double> gg(2 * DIM);
    for (int i = 0; i < win_size - 1; i++) {
      jtj.setZero();
      std::fill(gg.begin(), gg.end(), 0.0);
      residual += imus_factor[i]->give_evaluate(x_stats[i], x_stats[i + 1], jtj, gg, true);
      for (int c = 0; c < 2 * DIM; c++)
        for (int r = 0; r < 2 * DIM; r++) Hess(i * DIM + r, i * DIM + c) += jtj(r, c);
      for (int r = 0; r < 2 * DIM; r++) JacT[i * DIM + r] += gg[r];
    }
    for (double& h : Hess.a) h *= imu_coef;
    for (double& j : JacT) j *= imu_coef;
    residual *= (imu_coef * 0.5);

    for (int i = 0; i < tthd_num; i++) {
      if (i != 0) mthreads[i]->join();
      else voxhess.acc_evaluate2(poses, 0, (int)part, hessians[0], jacobins[0], resis[0]);
      hess_plus(Hess, JacT, hessians[i], jacobins[i]);
      residual += resis[i];
      delete mthreads[i];
    }
    return residual;
  }

  // :525-560
  double only_residual(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor) {
    double residual1 = 0, residual2 = 0;
    MatX jtj(2 * DIM, 2 * DIM);
    std::vector<double> gg(2 * DIM);
    int tn = thd_num;
    const int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tn) tn = 1;
    std::vector<double> residuals(tn, 0);
    std::vector<std::thread*> mthreads(tn, nullptr);
    const double part = 1.0 * g_size / tn;
    const std::vector<Pose> poses = poses_of(x_stats);
    for (int i = 1; i < tn; i++)
      mthreads[i] = new std::thread(&LidarFactor::evaluate_only_residual, &voxhess, poses, (int)(part * i), (int)(part * (i + 1)),
                                    std::ref(residuals[i]));
    for (int i = 0; i < win_size - 1; i++) residual1 += imus_factor[i]->give_evaluate(x_stats[i], x_stats[i + 1], jtj, gg, false);
    residual1 *= (imu_coef * 0.5);
    for (int i = 0; i < tn; i++) {
      if (i != 0) { mthreads[i]->join(); delete mthreads[i]; }
      else voxhess.evaluate_only_residual(poses, (int)(part * i), (int)(part * (i + 1)), residuals[i]);
      residual2 += residuals[i];
    }
    return residual1 + residual2;
  }

  // :562-653 (three iterations upstream; max_iter is a parameter here so tests can run longer schedules)
  void damping_iter(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, MatX* hess, int max_iter = 3) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    imu_leng = win_size * DIM;
    trace.clear();
    double u = 0.01, v = 2;
    MatX D(imu_leng, imu_leng), Hess(imu_leng, imu_leng);
    std::vector<double> JacT(imu_leng), dxi(imu_leng);
    hess->resize(imu_leng, imu_leng);
    double residual1 = 0, residual2 = 0, q;
    bool is_calc_hess = true;
    std::vector<ImuState> x_stats_temp = x_stats;

    for (int i = 0; i < max_iter; i++) {
      LMTraceEntry te{};
      te.recomputed_hess = is_calc_hess;
      if (is_calc_hess) {
        residual1 = divide_thread(x_stats, voxhess, imus_factor, Hess, JacT);
        *hess = Hess;
      }
      for (int r = 0; r < DIM; r++) for (int c = 0; c < imu_leng; c++) Hess(r, c) = 0.0;
      for (int c = 0; c < DIM; c++) for (int r = 0; r < imu_leng; r++) Hess(r, c) = 0.0;
      for (int r = 0; r < DIM; r++) Hess(r, r) = 1.0;
      for (int r = 0; r < DIM; r++) JacT[r] = 0.0;

      for (int r = 0; r < imu_leng; r++) D(r, r) = Hess(r, r);
      MatX A(imu_leng, imu_leng);
      for (int c = 0; c < imu_leng; c++)
        for (int r = 0; r < imu_leng; r++) A(r, c) = Hess(r, c) + (r == c ? u * D(r, r) : 0.0);
      std::vector<double> rhs(imu_leng);
      for (int r = 0; r < imu_leng; r++) rhs[r] = -JacT[r];
      dxi = ldlt_solve(A, rhs);

      for (int j = 0; j < win_size; j++) {
        const double* d = &dxi[DIM * j];
        x_stats_temp[j].R = x_stats[j].R * Exp(v3(d[0], d[1], d[2]));
        x_stats_temp[j].p = x_stats[j].p + v3(d[3], d[4], d[5]);
        x_stats_temp[j].v = x_stats[j].v + v3(d[6], d[7], d[8]);
        x_stats_temp[j].bg = x_stats[j].bg + v3(d[9], d[10], d[11]);
        x_stats_temp[j].ba = x_stats[j].ba + v3(d[12], d[13], d[14]);
      }
      for (int j = 0; j < win_size - 1; j++) imus_factor[j]->update_state(&dxi[DIM * j]);

      double q1 = 0;
      for (int r = 0; r < imu_leng; r++) q1 += dxi[r] * (u * D(r, r) * dxi[r] - JacT[r]);
      q1 *= 0.5;

      residual2 = only_residual(x_stats_temp, voxhess, imus_factor);
      q = residual1 - residual2;
      te.residual1 = residual1; te.residual2 = residual2; te.u = u; te.v = v; te.q = q; te.q1 = q1;

      if (q > 0) {
        x_stats = x_stats_temp;
        const double one_three = 1.0 / 3;
        q = q / q1;
        v = 2;
        q = 1 - std::pow(2 * q - 1, 3);
        u *= (q < one_three ? one_three : q);
        is_calc_hess = true;
        te.accepted = 1;
      } else {
        u = u * v;
        v = 2 * v;
        is_calc_hess = false;
        te.accepted = 0;
        for (int j = 0; j < win_size - 1; j++) {
          imus_factor[j]->dbg = imus_factor[j]->dbg_buf;
          imus_factor[j]->dba = imus_factor[j]->dba_buf;
        }
      }
      trace.push_back(te);
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
  }
};

// voxel_map.hpp:658-864.  Three gravity unknowns appended at the tail of the 15W system; only frame 0's POSE (6) is
// gauge-fixed (:800-803); x_stats_temp is never reset from x_stats, so the gravity of a rejected trial is kept and the
// next increment is added on top of it (:813) -- reproduced as is.
class LI_BA_OptimizerGravity {
 public:
  int win_size = 0, jac_leng = 0, imu_leng = 0, thd_num = 5;
  double imu_coef = 1e-4;
  std::vector<LMTraceEntry> trace;

  double divide_thread(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, MatX& Hess,
                       std::vector<double>& JacT) {
    double residual = 0;
    Hess.setZero();
    std::fill(JacT.begin(), JacT.end(), 0.0);
    std::vector<MatX> hessians(thd_num);
    std::vector<std::vector<double>> jacobins(thd_num);
    std::vector<double> resis(thd_num, 0);
    for (int i = 0; i < thd_num; i++) { hessians[i].resize(jac_leng, jac_leng); jacobins[i].assign(jac_leng, 0.0); }
    int tthd_num = thd_num;
    const int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tthd_num) tthd_num = 1;
    const double part = 1.0 * g_size / tthd_num;
    const std::vector<Pose> poses = LI_BA_Optimizer::poses_of(x_stats);
    std::vector<std::thread*> mthreads(tthd_num, nullptr);
    for (int i = 1; i < tthd_num; i++)
      mthreads[i] = new std::thread(&LidarFactor::acc_evaluate2, &voxhess, poses, (int)(part * i), (int)(part * (i + 1)),
                                    std::ref(hessians[i]), std::ref(jacobins[i]), std::ref(resis[i]));
    MatX jtj(2 * DIM + 3, 2 * DIM + 3);
    std::vector<double> gg(2 * DIM + 3);
    const int gq = imu_leng - 3;
    for (int i = 0; i < win_size - 1; i++) {
      residual += imus_factor[i]->give_evaluate_g(x_stats[i], x_stats[i + 1], jtj, gg, true);
      for (int c = 0; c < 2 * DIM; c++)
        for (int r = 0; r < 2 * DIM; r++) Hess(i * DIM + r, i * DIM + c) += jtj(r, c);
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 2 * DIM; r++) { Hess(i * DIM + r, gq + c) += jtj(r, 2 * DIM + c); Hess(gq + c, i * DIM + r) += jtj(2 * DIM + c, r); }
      for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) Hess(gq + r, gq + c) += jtj(2 * DIM + r, 2 * DIM + c);
      for (int r = 0; r < 2 * DIM; r++) JacT[i * DIM + r] += gg[r];
      for (int r = 0; r < 3; r++) JacT[gq + r] += gg[2 * DIM + r];
    }
    for (double& h : Hess.a) h *= imu_coef;
    for (double& j : JacT) j *= imu_coef;
    residual *= (imu_coef * 0.5);
    for (int i = 0; i < tthd_num; i++) {
      if (i != 0) mthreads[i]->join();
      else voxhess.acc_evaluate2(poses, 0, (int)part, hessians[0], jacobins[0], resis[0]);
      for (int a = 0; a < win_size; a++) {     // hess_plus :663-671
        for (int k = 0; k < DVEL; k++) JacT[a * DIM + k] += jacobins[i][a * DVEL + k];
        for (int b = 0; b < win_size; b++)
          for (int c = 0; c < DVEL; c++)
            for (int r = 0; r < DVEL; r++) Hess(a * DIM + r, b * DIM + c) += hessians[i](a * DVEL + r, b * DVEL + c);
      }
      residual += resis[i];
      delete mthreads[i];
    }
    return residual;
  }

  double only_residual(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor) {
    LI_BA_Optimizer o;   // same arithmetic: give_evaluate_g without Jacobian == give_evaluate without Jacobian (:738-773)
    o.win_size = win_size; o.jac_leng = jac_leng; o.imu_leng = imu_leng; o.thd_num = thd_num; o.imu_coef = imu_coef;
    return o.only_residual(x_stats, voxhess, imus_factor);
  }

  void damping_iter(std::vector<ImuState>& x_stats, LidarFactor& voxhess, std::deque<IMU_PRE*>& imus_factor, std::vector<double>& resis,
                    MatX* hess, int max_iter = 2) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    imu_leng = win_size * DIM + 3;
    trace.clear();
    double u = 0.01, v = 2;
    MatX D(imu_leng, imu_leng), Hess(imu_leng, imu_leng);
    std::vector<double> JacT(imu_leng), dxi(imu_leng);
    double residual1 = 0, residual2 = 0, q;
    bool is_calc_hess = true;
    std::vector<ImuState> x_stats_temp = x_stats;
    for (int i = 0; i < max_iter; i++) {
      LMTraceEntry te{};
      te.recomputed_hess = is_calc_hess;
      if (is_calc_hess) {
        residual1 = divide_thread(x_stats, voxhess, imus_factor, Hess, JacT);
        *hess = Hess;
      }
      if (i == 0) resis.push_back(residual1);
      for (int r = 0; r < 6; r++) for (int c = 0; c < imu_leng; c++) Hess(r, c) = 0.0;
      for (int c = 0; c < 6; c++) for (int r = 0; r < imu_leng; r++) Hess(r, c) = 0.0;
      for (int r = 0; r < 6; r++) { Hess(r, r) = 1.0; JacT[r] = 0.0; }
      for (int r = 0; r < imu_leng; r++) D(r, r) = Hess(r, r);
      MatX A(imu_leng, imu_leng);
      for (int c = 0; c < imu_leng; c++)
        for (int r = 0; r < imu_leng; r++) A(r, c) = Hess(r, c) + (r == c ? u * D(r, r) : 0.0);
      std::vector<double> rhs(imu_leng);
      for (int r = 0; r < imu_leng; r++) rhs[r] = -JacT[r];
      dxi = ldlt_solve(A, rhs);

      x_stats_temp[0].g = x_stats_temp[0].g + v3(dxi[imu_leng - 3], dxi[imu_leng - 2], dxi[imu_leng - 1]);
      for (int j = 0; j < win_size; j++) {
        const double* d = &dxi[DIM * j];
        x_stats_temp[j].R = x_stats[j].R * Exp(v3(d[0], d[1], d[2]));
        x_stats_temp[j].p = x_stats[j].p + v3(d[3], d[4], d[5]);
        x_stats_temp[j].v = x_stats[j].v + v3(d[6], d[7], d[8]);
        x_stats_temp[j].bg = x_stats[j].bg + v3(d[9], d[10], d[11]);
        x_stats_temp[j].ba = x_stats[j].ba + v3(d[12], d[13], d[14]);
        x_stats_temp[j].g = x_stats_temp[0].g;
      }
      for (int j = 0; j < win_size - 1; j++) imus_factor[j]->update_state(&dxi[DIM * j]);
      double q1 = 0;
      for (int r = 0; r < imu_leng; r++) q1 += dxi[r] * (u * D(r, r) * dxi[r] - JacT[r]);
      q1 *= 0.5;
      residual2 = only_residual(x_stats_temp, voxhess, imus_factor);
      q = residual1 - residual2;
      te.residual1 = residual1; te.residual2 = residual2; te.u = u; te.v = v; te.q = q; te.q1 = q1;
      if (q > 0) {
        x_stats = x_stats_temp;
        const double one_three = 1.0 / 3;
        q = q / q1;
        v = 2;
        q = 1 - std::pow(2 * q - 1, 3);
        u *= (q < one_three ? one_three : q);
        is_calc_hess = true;
        te.accepted = 1;
      } else {
        u = u * v;
        v = 2 * v;
        is_calc_hess = false;
        te.accepted = 0;
        for (int j = 0; j < win_size - 1; j++) { imus_factor[j]->dbg = imus_factor[j]->dbg_buf; imus_factor[j]->dba = imus_factor[j]->dba_buf; }
      }
      trace.push_back(te);
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    resis.push_back(residual2);
  }
};

}  // namespace vxo
