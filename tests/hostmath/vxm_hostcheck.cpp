// Test harness: compiles the DEVICE arithmetic header (voxel-slam_amd/csrc/vxba_math.hpp) with the
// host compiler and assembles Hess/JacT/residual and the K2 cache with plain loops, so the per-lane
// math of the HIP kernels can be checked against the oracle without a GPU.  Not part of the product.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../voxel-slam_amd/csrc/vxba_math.hpp"

extern "C" {

void vxmh_eig_sym3(const double* C6, double* lam, double* U_rowmajor) { vxm::eig_sym3(C6, lam, U_rowmajor); }
void vxmh_eig_sym3_warm(const double* C6, const double* Up_rowmajor, double* lam, double* U_rowmajor) { vxm::eig_sym3_warm(C6, Up_rowmajor, lam, U_rowmajor); }

static void pose_rowmajor(const double* Rp, double R[9], double p[3]) {
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) R[3 * r + c] = Rp[3 * c + r];
  for (int k = 0; k < 3; k++) p[k] = Rp[9 + k];
}

// K2 on the host: clusters V*W*10, fix V*10, coe V -> eig_val V*3, eig_vec V*9 (col-major), merged V*10, residual
void vxmh_k2(int V, int W, const double* clusters, const double* fix, const double* coe, const double* Rp, double* eig_val,
             double* eig_vec, double* merged, double* residual) {
  double res = 0;
  for (int a = 0; a < V; a++) {
    double SP[6], Sv[3], SN;
    for (int k = 0; k < 6; k++) SP[k] = fix[10 * a + k];
    for (int k = 0; k < 3; k++) Sv[k] = fix[10 * a + 6 + k];
    SN = fix[10 * a + 9];
    for (int i = 0; i < W; i++) {
      const double* c = clusters + ((size_t)a * W + i) * 10;
      if (c[9] == 0) continue;
      double R[9], p[3];
      pose_rowmajor(Rp + 12 * i, R, p);
      vxm::transform_accumulate(c, c + 6, c[9], R, p, SP, Sv, SN);
    }
    double C[6], lam[3], U[9];
    vxm::cluster_cov(SP, Sv, SN, C);
    vxm::eig_sym3(C, lam, U);
    for (int k = 0; k < 3; k++) eig_val[3 * a + k] = lam[k];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) eig_vec[9 * a + 3 * c + r] = U[3 * r + c];
    for (int k = 0; k < 6; k++) merged[10 * a + k] = SP[k];
    for (int k = 0; k < 3; k++) merged[10 * a + 6 + k] = Sv[k];
    merged[10 * a + 9] = SN;
    res += coe[a] * lam[0];
  }
  *residual = res;
}

// K2 with the f32 re-centred cluster records (VXBA_PRECISION_MIXED_F32_CLUSTERS): records out (V*W*10 floats), same outputs as vxmh_k2
void vxmh_k2_f32(int V, int W, const double* clusters, const double* fix, const double* coe, const double* Rp, float* records, double* eig_val,
                 double* merged, double* residual) {
  double res = 0;
  for (int a = 0; a < V; a++) {
    double SP[6], Sv[3], SN;
    for (int k = 0; k < 6; k++) SP[k] = fix[10 * a + k];
    for (int k = 0; k < 3; k++) Sv[k] = fix[10 * a + 6 + k];
    SN = fix[10 * a + 9];
    for (int i = 0; i < W; i++) {
      float* rec = records + ((size_t)a * W + i) * 10;
      vxm::cluster_to_centred_f32(clusters + ((size_t)a * W + i) * 10, rec);
      double R[9], p[3];
      pose_rowmajor(Rp + 12 * i, R, p);
      vxm::transform_accumulate_centred(rec, R, p, SP, Sv, SN);     // unobserved slots are all-zero records: no test, as in the kernel
    }
    double C[6], lam[3], U[9];
    vxm::cluster_cov(SP, Sv, SN, C);
    vxm::eig_sym3(C, lam, U);
    for (int k = 0; k < 3; k++) eig_val[3 * a + k] = lam[k];
    for (int k = 0; k < 6; k++) merged[10 * a + k] = SP[k];
    for (int k = 0; k < 3; k++) merged[10 * a + 6 + k] = Sv[k];
    merged[10 * a + 9] = SN;
    res += coe[a] * lam[0];
  }
  *residual = res;
}

// K3 on the host via the rank-3 rows: Hess (6W)^2 col-major, JacT 6W, residual
// spare != 0: the narrow-window kernel's variant -- k3_entry<false> leaves Drt / Dtt out of the accumulators and the z row of
// every voxel carries sqrt2 sqrt(coe) u in three extra columns, whose products with the frame columns ARE Drt / Dtt
// (vxba_k3.hpp, K3Cfg::SPARE).
static void vxmh_k3_impl(int V, int W, const double* clusters, const double* coe, const double* eig_val, const double* eig_vec,
             const double* merged, const double* Rp, double* Hess, double* JacT, double* residual, int spare);
void vxmh_k3(int V, int W, const double* clusters, const double* coe, const double* eig_val, const double* eig_vec,
             const double* merged, const double* Rp, double* Hess, double* JacT, double* residual) {
  vxmh_k3_impl(V, W, clusters, coe, eig_val, eig_vec, merged, Rp, Hess, JacT, residual, 0);
}
void vxmh_k3_spare(int V, int W, const double* clusters, const double* coe, const double* eig_val, const double* eig_vec,
             const double* merged, const double* Rp, double* Hess, double* JacT, double* residual) {
  vxmh_k3_impl(V, W, clusters, coe, eig_val, eig_vec, merged, Rp, Hess, JacT, residual, 1);
}
static void vxmh_k3_impl(int V, int W, const double* clusters, const double* coe, const double* eig_val, const double* eig_vec,
             const double* merged, const double* Rp, double* Hess, double* JacT, double* residual, int spare) {
  const int n = 6 * W;
  std::vector<double> SP((size_t)n * 3, 0.0);           // S[x][6W + k]: products of the z row with the spare columns
  std::vector<double> S((size_t)n * n, 0.0);            // sum rows^T rows (row-major)
  std::vector<double> acc((size_t)W * 27, 0.0);
  double res = 0;
  for (int a = 0; a < V; a++) {
    vxm::VoxelCache vc;
    const double* U = eig_vec + 9 * a;
    for (int k = 0; k < 3; k++) { vc.u0[k] = U[k]; vc.u1[k] = U[3 + k]; vc.u2[k] = U[6 + k]; }
    vxm::gap_scales(eig_val + 3 * a, vc.s1, vc.s2);
    vc.invN = 1.0 / merged[10 * a + 9];
    for (int k = 0; k < 3; k++) vc.vbar[k] = merged[10 * a + 6 + k] * vc.invN;
    vc.coe = coe[a];
    vc.sc = std::sqrt(coe[a]);
    std::vector<double> B((size_t)3 * n, 0.0);
    for (int i = 0; i < W; i++) {
      const double* c = clusters + ((size_t)a * W + i) * 10;
      if (c[9] == 0) continue;
      double R[9], p[3], rows[3][6];
      pose_rowmajor(Rp + 12 * i, R, p);
      if (spare) vxm::k3_entry<false>(c, c + 6, c[9], R, p, vc, rows, acc.data() + 27 * i);
      else vxm::k3_entry(c, c + 6, c[9], R, p, vc, rows, acc.data() + 27 * i);
      for (int r = 0; r < 3; r++) for (int k = 0; k < 6; k++) B[(size_t)r * n + 6 * i + k] = rows[r][k];
    }
    if (spare) {
      const double ss = 1.4142135623730951 * vc.sc;
      for (int x = 0; x < n; x++) for (int k = 0; k < 3; k++) SP[(size_t)x * 3 + k] += B[(size_t)2 * n + x] * (ss * vc.u0[k]);
    }
    for (int r = 0; r < 3; r++)
      for (int x = 0; x < n; x++) {
        double bx = B[(size_t)r * n + x];
        if (bx == 0) continue;
        for (int y = 0; y < n; y++) S[(size_t)x * n + y] += bx * B[(size_t)r * n + y];
      }
    res += coe[a] * eig_val[3 * a];
  }
  for (int x = 0; x < n; x++) for (int y = 0; y < n; y++) Hess[(size_t)y * n + x] = -S[(size_t)x * n + y];
  const int s6i[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  for (int i = 0; i < W; i++) {
    const double* d = acc.data() + 27 * i;
    for (int k = 0; k < 6; k++) JacT[6 * i + k] = d[k];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        Hess[(size_t)(6 * i + c) * n + 6 * i + r] += d[6 + s6i[r][c]];
        Hess[(size_t)(6 * i + 3 + c) * n + 6 * i + r] += d[12 + 3 * r + c];      // (rot r, trans c)
        Hess[(size_t)(6 * i + r) * n + 6 * i + 3 + c] += d[12 + 3 * r + c];      // transpose
        Hess[(size_t)(6 * i + 3 + c) * n + 6 * i + 3 + r] += d[21 + s6i[r][c]];
      }
    if (spare)   // k3_finalize's rule: Hess(6i+a, 6i+b) += S[6i+a][6W + (b-3)] for b >= 3, a <= b, mirrored
      for (int a = 0; a < 6; a++)
        for (int b = (a < 3 ? 3 : a); b < 6; b++) {
          const double v = SP[(size_t)(6 * i + a) * 3 + (b - 3)];
          Hess[(size_t)(6 * i + b) * n + 6 * i + a] += v;
          if (a != b) Hess[(size_t)(6 * i + a) * n + 6 * i + b] += v;
        }
  }
  *residual = res;
}
}
