// TEST INFRASTRUCTURE ONLY (see vxo_linalg.hpp header).  PINNED against the reference's own code compiled here (oracle/_ref/libref.so, `make -C oracle ref`: tests/test_ref_pin.py):
// PointCluster, LidarFactor (acc_evaluate2, evaluate_only_residual) and Lidar_BA_Optimizer::damping_iter incl. rejected steps -- through the
// unmodified voxel_map.hpp / tools.hpp.  Eigen is absent: SelfAdjointEigenSolver / LDLT inside libref are this directory's restatements
// (shim/Eigen), so for those three algorithms the pin is circular (vxo_linalg.hpp header).
//
// CPU restatement of the reference's LiDAR BA factor and LM optimizer:
//   PointCluster                 <- VoxelSLAM/src/tools.hpp:304-365
//   Pose (R,p of IMUST)          <- VoxelSLAM/src/tools.hpp:135-199
//   LidarFactor                  <- VoxelSLAM/src/voxel_map.hpp:109-290
//   Lidar_BA_Optimizer           <- VoxelSLAM/src/voxel_map.hpp:293-444
// Same arithmetic, same loop order, same thread fan-out (std::thread over
// contiguous voxel ranges, private accumulators, serial sum in thread order), so
// the same object doubles as the timed "reference-equivalent" CPU baseline.
#pragma once
#include <cstdio>
#include <functional>
#include <thread>
#include <vector>

#include "vxo_linalg.hpp"

namespace vxo {

// tools.hpp:304-365
struct PointCluster {
  M3 P;
  V3 v;
  int N;
  PointCluster() { clear(); }
  void clear() { P = zero33(); v = zero3(); N = 0; }
  // tools.hpp:326-331
  void push(const V3& vec) { N++; P = P + outer(vec, vec); v = v + vec; }
  // tools.hpp:333-337
  M3 cov() const { V3 center = v / (double)N; return P / (double)N - outer(center, center); }
  // tools.hpp:339-355
  PointCluster& operator+=(const PointCluster& s) { P = P + s.P; v = v + s.v; N += s.N; return *this; }
  PointCluster& operator-=(const PointCluster& s) { P = P - s.P; v = v - s.v; N -= s.N; return *this; }
};

// The part of IMUST the LiDAR factor reads (tools.hpp:139-140).
struct Pose {
  M3 R;
  V3 p;
};

// tools.hpp:357-363
inline void cluster_transform(PointCluster& out, const PointCluster& s, const Pose& x) {
  out.N = s.N;
  out.v = x.R * s.v + (double)out.N * x.p;
  M3 rp = outer(x.R * s.v, x.p);
  out.P = x.R * s.P * transpose(x.R) + rp + transpose(rp) + outer((double)out.N * x.p, x.p);
}

// voxel_map.hpp:109-290
class LidarFactor {
 public:
  std::vector<PointCluster> sig_vecs;
  std::vector<std::vector<PointCluster>> plvec_voxels;
  std::vector<double> coeffs;
  std::vector<V3> eig_values;
  std::vector<M3> eig_vectors;
  std::vector<PointCluster> pcr_adds;
  int win_size;

  explicit LidarFactor(int w) : win_size(w) {}

  // voxel_map.hpp:122-130
  void push_voxel(const std::vector<PointCluster>& vec_orig, const PointCluster& fix, double coe,
                  const V3& eig_value, const M3& eig_vector, const PointCluster& pcr_add) {
    plvec_voxels.push_back(vec_orig);
    sig_vecs.push_back(fix);
    coeffs.push_back(coe);
    eig_values.push_back(eig_value);
    eig_vectors.push_back(eig_vector);
    pcr_adds.push_back(pcr_add);
  }

  // voxel_map.hpp:132-241.  Uses the CACHED (lambda, U, merged) -- it never
  // recomputes the eigen-decomposition (the recompute is commented out upstream).
  void acc_evaluate2(const std::vector<Pose>& xs, int head, int end, MatX& Hess, std::vector<double>& JacT,
                     double& residual) const {
    Hess.setZero();
    std::fill(JacT.begin(), JacT.end(), 0.0);
    residual = 0;
    const int kk = 0;
    std::vector<V3> viRiTuk(win_size);
    std::vector<M3> viRiTukukT(win_size);
    struct M36 { double a[3][6]; };
    std::vector<M36> Auk(win_size);

    for (int a = head; a < end; a++) {
      const std::vector<PointCluster>& sig_orig = plvec_voxels[a];
      double coe = coeffs[a];

      V3 lmbd = eig_values[a];
      M3 U = eig_vectors[a];
      int NN = pcr_adds[a].N;
      V3 vBar = pcr_adds[a].v / (double)NN;

      V3 u[3] = {col(U, 0), col(U, 1), col(U, 2)};
      const V3& uk = u[kk];
      M3 ukukT = outer(uk, uk);
      M3 umumT = zero33();
      for (int i = 0; i < 3; i++)
        if (i != kk) umumT = umumT + outer((2.0 / (lmbd[kk] - lmbd[i])) * u[i], u[i]);   // (s * u_i) * u_i^T, the reference's association (:172)

      for (int i = 0; i < win_size; i++)
        if (sig_orig[i].N != 0) {
          const M3& Pi = sig_orig[i].P;
          const V3& vi = sig_orig[i].v;
          const M3& Ri = xs[i].R;
          double ni = sig_orig[i].N;

          M3 vihat = hat(vi);
          V3 RiTuk = transpose(Ri) * uk;
          M3 RiTukhat = hat(RiTuk);

          V3 PiRiTuk = Pi * RiTuk;
          viRiTuk[i] = vihat * RiTuk;
          viRiTukukT[i] = outer(viRiTuk[i], uk);

          V3 ti_v = xs[i].p - vBar;
          double ukTti_v = dot(uk, ti_v);

          M3 combo1 = hat(PiRiTuk) + vihat * ukTti_v;
          V3 combo2 = Ri * vi + ni * ti_v;
          M3 left = (Ri * Pi + outer(ti_v, vi)) * RiTukhat - Ri * combo1;
          M3 right = outer(combo2, uk) + dot(combo2, uk) * eye33();
          for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
              Auk[i].a[r][c] = left(r, c) / NN;
              Auk[i].a[r][c + 3] = right(r, c) / NN;
            }

          double jjt[6];
          for (int c = 0; c < 6; c++) jjt[c] = Auk[i].a[0][c] * uk[0] + Auk[i].a[1][c] * uk[1] + Auk[i].a[2][c] * uk[2];
          for (int c = 0; c < 6; c++) JacT[6 * i + c] += coe * jjt[c];

          M3 HRt = (2.0 / NN * (1.0 - ni / NN)) * viRiTukukT[i];
          double Hb[6][6];
          aT_m_b(Auk[i].a, umumT, Auk[i].a, Hb);
          // :208, associated as the expression text is: ((2/NN) * (..)) * RiTukhat  and  ((2/NN/NN) * v) * v^T
          M3 blk = ((2.0 / NN) * (combo1 - RiTukhat * Pi)) * RiTukhat - outer((2.0 / NN / NN) * viRiTuk[i], viRiTuk[i]) -
                   0.5 * hat(v3(jjt[0], jjt[1], jjt[2]));
          for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++) {
              Hb[r][c] += blk(r, c);
              Hb[r][c + 3] += HRt(r, c);
              Hb[r + 3][c] += HRt(c, r);
              Hb[r + 3][c + 3] += 2.0 / NN * (ni - ni * ni / NN) * ukukT(r, c);
            }
          for (int r = 0; r < 6; r++)
            for (int c = 0; c < 6; c++) Hess(6 * i + r, 6 * i + c) += coe * Hb[r][c];
        }

      for (int i = 0; i < win_size - 1; i++)
        if (sig_orig[i].N != 0) {
          double ni = sig_orig[i].N;
          for (int j = i + 1; j < win_size; j++)
            if (sig_orig[j].N != 0) {
              double nj = sig_orig[j].N;
              double Hb[6][6];
              aT_m_b(Auk[i].a, umumT, Auk[j].a, Hb);
              M3 ww = outer((-2.0 / NN / NN) * viRiTuk[i], viRiTuk[j]);   // ((-2/NN/NN) * v_i) * v_j^T (:225)
              for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                  Hb[r][c] += ww(r, c);
                  Hb[r][c + 3] += -2.0 * nj / NN / NN * viRiTukukT[i](r, c);
                  Hb[r + 3][c] += -2.0 * ni / NN / NN * viRiTukukT[j](c, r);
                  Hb[r + 3][c + 3] += -2.0 * ni * nj / NN / NN * ukukT(r, c);
                }
              for (int r = 0; r < 6; r++)
                for (int c = 0; c < 6; c++) Hess(6 * i + r, 6 * j + c) += coe * Hb[r][c];
            }
        }

      residual += coe * lmbd[kk];
    }

    // voxel_map.hpp:237-239: mirror the upper block triangle
    for (int i = 1; i < win_size; i++)
      for (int j = 0; j < i; j++)
        for (int r = 0; r < 6; r++)
          for (int c = 0; c < 6; c++) Hess(6 * i + r, 6 * j + c) = Hess(6 * j + c, 6 * i + r);
  }

  // voxel_map.hpp:243-279.  Writes the (lambda, U, merged) cache.
  void evaluate_only_residual(const std::vector<Pose>& xs, int head, int end, double& residual) {
    residual = 0;
    const int kk = 0;
    PointCluster pcr;
    for (int a = head; a < end; a++) {
      const std::vector<PointCluster>& sig_orig = plvec_voxels[a];
      PointCluster sig = sig_vecs[a];
      for (int i = 0; i < win_size; i++)
        if (sig_orig[i].N != 0) {
          cluster_transform(pcr, sig_orig[i], xs[i]);
          sig += pcr;
        }
      V3 vBar = sig.v / (double)sig.N;
      M3 C = sig.P / (double)sig.N - outer(vBar, vBar);
      V3 lmbd; M3 U;
      eig_sym3(C, lmbd, U);
      eig_values[a] = lmbd;
      eig_vectors[a] = U;
      pcr_adds[a] = sig;
      residual += coeffs[a] * lmbd[kk];
    }
  }

  // voxel_map.hpp:281-286
  void clear() {
    sig_vecs.clear(); plvec_voxels.clear();
    eig_values.clear(); eig_vectors.clear();
    pcr_adds.clear(); coeffs.clear();
  }

 private:
  // out(6x6) = A^T (3x6)^T * M (3x3) * B (3x6), evaluated as (A^T M) B like the Eigen expression
  static void aT_m_b(const double A[3][6], const M3& M, const double B[3][6], double out[6][6]) {
    double AtM[6][3];
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 3; c++) AtM[r][c] = A[0][r] * M(0, c) + A[1][r] * M(1, c) + A[2][r] * M(2, c);
    for (int r = 0; r < 6; r++)
      for (int c = 0; c < 6; c++) out[r][c] = AtM[r][0] * B[0][c] + AtM[r][1] * B[1][c] + AtM[r][2] * B[2][c];
  }
};

struct LMTraceEntry {
  double residual1, residual2, u, v, q, q1;
  int accepted, recomputed_hess;
};

// voxel_map.hpp:293-444
class Lidar_BA_Optimizer {
 public:
  int win_size = 0, jac_leng = 0, thd_num = 2;
  std::vector<LMTraceEntry> trace;

  // voxel_map.hpp:298-335
  double divide_thread(std::vector<Pose>& x_stats, LidarFactor& voxhess, MatX& Hess, std::vector<double>& JacT) {
    double residual = 0;
    Hess.setZero();
    std::fill(JacT.begin(), JacT.end(), 0.0);
    std::vector<MatX> hessians(thd_num);
    std::vector<std::vector<double>> jacobins(thd_num);
    for (int i = 0; i < thd_num; i++) {
      hessians[i].resize(jac_leng, jac_leng);
      jacobins[i].assign(jac_leng, 0.0);
    }
    int tthd_num = thd_num;
    std::vector<double> resis(tthd_num, 0);
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < tthd_num) tthd_num = 1;

    std::vector<std::thread*> mthreads(tthd_num, nullptr);
    double part = 1.0 * g_size / tthd_num;
    for (int i = 1; i < tthd_num; i++)
      mthreads[i] = new std::thread(&LidarFactor::acc_evaluate2, &voxhess, x_stats, (int)(part * i), (int)(part * (i + 1)),
                                    std::ref(hessians[i]), std::ref(jacobins[i]), std::ref(resis[i]));
    for (int i = 0; i < tthd_num; i++) {
      if (i != 0) mthreads[i]->join();
      else voxhess.acc_evaluate2(x_stats, 0, (int)part, hessians[0], jacobins[0], resis[0]);
      for (size_t k = 0; k < Hess.a.size(); k++) Hess.a[k] += hessians[i].a[k];
      for (int k = 0; k < jac_leng; k++) JacT[k] += jacobins[i][k];
      residual += resis[i];
      delete mthreads[i];
    }
    return residual;
  }

  // voxel_map.hpp:337-365.  The reference exit(0)s when there are fewer voxels
  // than threads (:345-348); the oracle reports that as NaN instead of exiting.
  double only_residual(std::vector<Pose>& x_stats, LidarFactor& voxhess) {
    double residual1 = 0;
    std::vector<double> residuals(thd_num, 0);
    int g_size = (int)voxhess.plvec_voxels.size();
    if (g_size < thd_num) return std::numeric_limits<double>::quiet_NaN();
    std::vector<std::thread*> mthreads(thd_num, nullptr);
    double part = 1.0 * g_size / thd_num;
    for (int i = 1; i < thd_num; i++)
      mthreads[i] = new std::thread(&LidarFactor::evaluate_only_residual, &voxhess, x_stats, (int)(part * i),
                                    (int)(part * (i + 1)), std::ref(residuals[i]));
    for (int i = 0; i < thd_num; i++) {
      if (i != 0) mthreads[i]->join();
      else voxhess.evaluate_only_residual(x_stats, (int)(part * i), (int)(part * (i + 1)), residuals[i]);
      residual1 += residuals[i];
      delete mthreads[i];
    }
    return residual1;
  }

  // voxel_map.hpp:367-442
  bool damping_iter(std::vector<Pose>& x_stats, LidarFactor& voxhess, MatX* hess, std::vector<double>& resis,
                    int max_iter = 3) {
    win_size = voxhess.win_size;
    jac_leng = win_size * 6;
    trace.clear();

    double u = 0.01, v = 2;
    MatX D(jac_leng, jac_leng), Hess(jac_leng, jac_leng);
    std::vector<double> JacT(jac_leng), dxi(jac_leng);
    hess->resize(jac_leng, jac_leng);
    for (int i = 0; i < jac_leng; i++) D(i, i) = 1.0;
    double residual1 = 0, residual2 = 0, q;
    bool is_calc_hess = true;
    std::vector<Pose> x_stats_temp = x_stats;
    bool is_converge = true;

    for (int i = 0; i < max_iter; i++) {
      LMTraceEntry te{};
      te.recomputed_hess = is_calc_hess;
      if (is_calc_hess) {
        residual1 = divide_thread(x_stats, voxhess, Hess, JacT);
        *hess = Hess;
      }
      if (i == 0) resis.push_back(residual1);

      // gauge fix on frame 0 (voxel_map.hpp:397-400)
      for (int r = 0; r < 6; r++) for (int c = 0; c < jac_leng; c++) Hess(r, c) = 0.0;
      for (int c = 0; c < 6; c++) for (int r = 0; r < jac_leng; r++) Hess(r, c) = 0.0;
      for (int r = 0; r < 6; r++) Hess(r, r) = 1.0;
      for (int r = 0; r < 6; r++) JacT[r] = 0.0;

      for (int r = 0; r < jac_leng; r++) D(r, r) = Hess(r, r);
      MatX A(jac_leng, jac_leng);
      for (int c = 0; c < jac_leng; c++)
        for (int r = 0; r < jac_leng; r++) A(r, c) = Hess(r, c) + u * D(r, c);
      std::vector<double> rhs(jac_leng);
      for (int r = 0; r < jac_leng; r++) rhs[r] = -JacT[r];
      dxi = ldlt_solve(A, rhs);

      for (int j = 0; j < win_size; j++) {
        x_stats_temp[j].R = x_stats[j].R * Exp(v3(dxi[6 * j], dxi[6 * j + 1], dxi[6 * j + 2]));
        x_stats_temp[j].p = x_stats[j].p + v3(dxi[6 * j + 3], dxi[6 * j + 4], dxi[6 * j + 5]);
      }
      double q1 = 0;
      for (int r = 0; r < jac_leng; r++) q1 += dxi[r] * (u * D(r, r) * dxi[r] - JacT[r]);
      q1 *= 0.5;

      residual2 = only_residual(x_stats_temp, voxhess);
      q = (residual1 - residual2);
      te.residual1 = residual1; te.residual2 = residual2; te.u = u; te.v = v; te.q = q; te.q1 = q1;

      if (q > 0) {
        x_stats = x_stats_temp;
        double one_three = 1.0 / 3;
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
        is_converge = false;
        te.accepted = 0;
      }
      trace.push_back(te);
      if (std::fabs((residual1 - residual2) / residual1) < 1e-6) break;
    }
    resis.push_back(residual2);
    return is_converge;
  }
};

}  // namespace vxo
