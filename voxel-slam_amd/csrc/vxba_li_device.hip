// Device-resident LiDAR-inertial LM loop (LI_BA_Optimizer::damping_iter, voxel_map.hpp:562-653): the inertial
// half of the joint system, the damped solve, the state update and the accept / reject step as kernels, so that one call enqueues
// all its iterations without a host round trip -- like the LiDAR-only loop of vxba_kernels.hip, whose sweeps it reuses in LM mode
// (poses and gating flags are mirrored into that loop's control block).
//
//   li_imu_kernel       one wave per IMU factor: residual + 15x30 Jacobian (lane 0, vxi::imu_residual_jac -- the code the host shell
//                       runs), then cov^-1 J, J^T cov^-1 J, J^T cov^-1 r spread over the wave        preintegration.hpp:137-212
//   li_assemble_kernel  one thread per entry of the (15W)^2 joint Hessian: imu_coef * IMU blocks + scattered LiDAR blocks
//                       (divide_thread + hess_plus, voxel_map.hpp:455-463, 493-521), stored in (pose | velocity-bias) block form
//   li_solve_kernel     ONE wave.  The unknowns split into poses x (6 per frame) and the rest y = (v, bg, ba) (9 per frame).  Only the
//                       IMU factors touch y, and they chain neighbouring frames, so C = H_yy is block-tridiagonal with 9x9 blocks:
//                       a block Thomas sweep gives Z = C^-1 [H_yx | g_y], the Schur complement S = H_xx - H_xy Z_x is a dense 6W system
//                       that goes through the one-wave elimination of the LiDAR-only loop, then dy = -z_g - Z_x dx.  Same step as the
//                       reference's pivoted LDL^T of the whole 15W system to round-off (the damped system is positive definite).
//   li_decide_kernel    residual of the trial state (voxel sweep partials + IMU), gain ratio, damping update, accept / roll back
//                       (voxel_map.hpp:611-647)
#include <hip/hip_runtime.h>

#include "vxba_imu.hpp"
#include "vxba_kernels.h"
#include "vxba_li_device.h"
#include "vxba_solve.hpp"

namespace vxli {

using vxk::LMState;
constexpr int DIM = 15, SL = 24, IL = 304;

// ---- IMU factors ------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void li_imu_kernel(LIState* __restrict__ li, int W, int trial) {
  __shared__ double J[DIM * 30], CJ[DIM * 30], CI[DIM * DIM], rr[DIM], q[DIM];
  if (li->done || (!trial && !li->calc_hess)) return;
  const int f = blockIdx.x, lane = threadIdx.x;
  const double* st = trial ? li->trial : li->states;
  const double* imu = li->imus + (size_t)IL * f;
  for (int e = lane; e < DIM * DIM; e += 64) CI[e] = li->cov_inv[(size_t)225 * f + e];
  if (lane == 0) vxi::imu_residual_jac(imu, st + SL * f, st + SL * (f + 1), !trial, false, rr, J);
  __syncthreads();
  if (lane < DIM) {   // q = cov^-1 r  (cov_inv[a*15+b] = element (b, a))
    double s = 0.0;
    for (int a = 0; a < DIM; a++) s += CI[a * DIM + lane] * rr[a];
    q[lane] = s;
  }
  __syncthreads();
  if (lane == 0) {
    double s = 0.0;
    for (int i = 0; i < DIM; i++) s += rr[i] * q[i];
    (trial ? li->imu_res_trial : li->imu_res)[f] = s;
  }
  if (trial) return;
  for (int e = lane; e < DIM * 30; e += 64) {      // CJ = cov^-1 J
    const int a = e % DIM, i = e / DIM;
    double s = 0.0;
    for (int b = 0; b < DIM; b++) s += CI[b * DIM + a] * J[i * DIM + b];
    CJ[e] = s;
  }
  __syncthreads();
  double* jtj = li->jtj + (size_t)900 * f;
  for (int e = lane; e < 900; e += 64) {           // jtj(i, j) = J(:, i) . CJ(:, j)
    const int i = e % 30, j = e / 30;
    double s = 0.0;
    for (int p = 0; p < DIM; p++) s += J[i * DIM + p] * CJ[j * DIM + p];
    jtj[e] = s;
  }
  if (lane < 30) {
    double s = 0.0;
    for (int b = 0; b < DIM; b++) s += J[lane * DIM + b] * q[b];
    li->gg[30 * f + lane] = s;
  }
}

// ---- joint system -------------------------------------------------------------------------------------------------------------
// entry (r, c) of the IMU part: sum over the factors that touch both frames (factor f couples frames f and f + 1)
__device__ inline double imu_entry(const LIState* li, int W, int fr, int lr, int fc, int lc) {
  double s = 0.0;
  if (fr == fc) {
    if (fr >= 1) s += li->jtj[(size_t)900 * (fr - 1) + (15 + lc) * 30 + 15 + lr];
    if (fr <= W - 2) s += li->jtj[(size_t)900 * fr + lc * 30 + lr];
  } else if (fc == fr + 1) {
    s = li->jtj[(size_t)900 * fr + (15 + lc) * 30 + lr];
  } else if (fr == fc + 1) {
    s = li->jtj[(size_t)900 * fc + lc * 30 + 15 + lr];
  }
  return s;
}

__global__ __launch_bounds__(256) void li_assemble_kernel(LIState* __restrict__ li, const double* __restrict__ packed, int W, double* __restrict__ hess_out) {
  if (li->done || !li->calc_hess) return;
  const int n = DIM * W, n6 = 6 * W;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const double coef = li->imu_coef;
  if (e < n * n) {
    const int r = e % n, c = e / n;
    const int fr = r / DIM, lr = r % DIM, fc = c / DIM, lc = c % DIM;
    const int df = fc - fr;
    double h = (df >= -1 && df <= 1) ? imu_entry(li, W, fr, lr, fc, lc) * coef : 0.0;
    if (lr < 6 && lc < 6) h += packed[(size_t)(6 * fc + lc) * n6 + 6 * fr + lr];
    if (hess_out) hess_out[e] = h;
    if (lr < 6 && lc < 6) li->Hxx[(size_t)(6 * fc + lc) * n6 + 6 * fr + lr] = h;
    else if (lr < 6 && df >= -1 && df <= 1) {
      li->B[((size_t)fr * 3 + df + 1) * 54 + (lc - 6) * 6 + lr] = h;
      li->R0[((size_t)fc * 9 + lc - 6) * (n6 + 1) + 6 * fr + lr] = fr >= 1 ? h : 0.0;   // H_yx(rest_fc, pose_fr) = H_xy^T; frame 0 is the gauge
    } else if (lr < 6) {
      li->R0[((size_t)fc * 9 + lc - 6) * (n6 + 1) + 6 * fr + lr] = 0.0;
    }
    else if (lr >= 6 && lc >= 6 && df == 0) li->Cd[(size_t)fr * 81 + (lc - 6) * 9 + lr - 6] = h;
    else if (lr >= 6 && lc >= 6 && df == 1) li->Co[(size_t)fr * 81 + (lc - 6) * 9 + lr - 6] = h;
  } else if (e < n * n + n) {
    const int r = e - n * n, fr = r / DIM, lr = r % DIM;
    double g = 0.0;
    if (fr >= 1) g += li->gg[30 * (fr - 1) + 15 + lr];
    if (fr <= W - 2) g += li->gg[30 * fr + lr];
    g *= coef;
    if (lr < 6) g += packed[(size_t)n6 * n6 + 6 * fr + lr];
    else li->R0[((size_t)fr * 9 + lr - 6) * (n6 + 1) + n6] = g;
    li->g[r] = g;
  } else if (e == n * n + n) {
    double s = 0.0;
    for (int f = 0; f < W - 1; f++) s += li->imu_res[f];
    li->residual1 = s * (coef * 0.5) + packed[(size_t)n6 * n6 + n6];
  }
}

// ---- the damped step ------------------------------------------------------------------------------------------------------------
template <int W>
__global__ __launch_bounds__(64) void li_solve_kernel(LIState* __restrict__ li, LMState* __restrict__ lm) {
  constexpr int n6 = 6 * W, NC = n6 + 1, NF = W > 1 ? W : 2, NT = (9 * NC + 63) / 64;
  __shared__ double colbuf[vxk::SOLVE_LDS];
  __shared__ double Rr[NF - 1][9][NC];   // frame j at index j - 1 (frame 0 is the gauge);    // right-hand sides [H_yx | g_y] of the block-tridiagonal solve, overwritten by Z = C^-1 [...]
  __shared__ double Ci[NF][81];       // inverses of the eliminated diagonal blocks
  __shared__ double Lk[81], Tm[81];
  __shared__ double Cds[81 * NF], Cos[81 * NF];   // C_jj and C_{j,j+1}
  __shared__ double xs[64], ys[9 * NF], gy[9 * NF];
  if (li->done) return;
  const int lane = threadIdx.x;
  const double u = li->u;
#define LI_STAMP(k) do { if (lane == 0) li->dbg[k] = clock64(); } while (0)
  LI_STAMP(0);
  // 1. right-hand sides [H_yx | g_y] (laid out by li_assemble_kernel) and the velocity-bias blocks: contiguous, all loads in flight at once
  {
    constexpr int TOT = 9 * NC * (W - 1), NL = (TOT + 63) / 64, BATCH = 16;
    const double* src = li->R0 + 9 * NC;   // frame 1 onwards
    double* dst = &Rr[0][0][0];
    for (int t0 = 0; t0 < NL; t0 += BATCH) {   // BATCH loads in flight, then BATCH LDS stores
      double v[BATCH];
#pragma unroll
      for (int t = 0; t < BATCH; t++) {
        const int e = lane + 64 * (t0 + t);
        v[t] = e < TOT ? src[e] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < BATCH; t++) {
        const int e = lane + 64 * (t0 + t);
        if (e < TOT) dst[e] = v[t];
      }
    }
    constexpr int TC = 81 * W, NLC = (TC + 63) / 64;
    double vd[NLC], vo[NLC];
#pragma unroll
    for (int t = 0; t < NLC; t++) {
      const int e = lane + 64 * t;
      vd[t] = e < TC ? li->Cd[e] : 0.0;
      vo[t] = e < TC ? li->Co[e] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < NLC; t++) {
      const int e = lane + 64 * t;
      if (e < TC) { Cds[e] = vd[t]; Cos[e] = vo[t]; }
    }
  }
  __builtin_amdgcn_wave_barrier();
  for (int e = lane; e < 9 * W; e += 64) gy[e] = e >= 9 ? Rr[e / 9 - 1][e % 9][n6] : 0.0;
  __builtin_amdgcn_wave_barrier();
  LI_STAMP(1);
  // 2. block Thomas, forward: T_j = C_jj (1 + u on the diagonal) - L_j C_{j-1,j},  L_j = C_{j,j-1} T_{j-1}^-1,  R_j -= L_j R_{j-1}
  for (int j = 1; j < W; j++) {
    const double* Cd = Cds + 81 * j;
    const double* Cp = Cos + 81 * (j - 1);   // C_{j-1,j}, column-major (rows: rest_{j-1})
    if (j > 1) {
      for (int e = lane; e < 81; e += 64) {
        const int a = e % 9, b = e / 9;
        double s = 0.0;
        for (int k = 0; k < 9; k++) s += Cp[a * 9 + k] * Ci[j - 1][b * 9 + k];   // C_{j,j-1}(a,k) = C_{j-1,j}(k,a)
        Lk[b * 9 + a] = s;
      }
      __builtin_amdgcn_wave_barrier();
    }
    for (int e = lane; e < 81; e += 64) {
      const int a = e % 9, b = e / 9;
      double t = Cd[e];
      if (a == b) t += u * t;
      if (j > 1)
        for (int k = 0; k < 9; k++) t -= Lk[k * 9 + a] * Cp[b * 9 + k];
      Tm[e] = t;
    }
    if (j > 1) {
#pragma unroll
      for (int t = 0; t < NT; t++) {     // every lane owns entries lane, lane + 64, ...: reads and writes only its own, plus the finished row block j - 1
        const int e = lane + 64 * t;
        if (e < 9 * NC) {
          const int a = e / NC, c = e % NC;
          double s = Rr[j - 1][a][c];
#pragma unroll
          for (int k = 0; k < 9; k++) s -= Lk[k * 9 + a] * Rr[j - 2][k][c];
          Rr[j - 1][a][c] = s;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // in-place Gauss-Jordan inverse of the 9x9 block (positive definite: no pivoting)
    for (int p = 0; p < 9; p++) {
      const double ip = vxk::fast_rcp_f64(Tm[p * 9 + p]);
      double nv[2] = {0.0, 0.0};
#pragma unroll
      for (int t = 0; t < 2; t++) {
        const int e = lane + 64 * t;
        if (e < 81) {
          const int r = e % 9, c = e / 9;
          double v;
          if (r == p && c == p) v = ip;
          else if (r == p) v = Tm[c * 9 + p] * ip;
          else if (c == p) v = -Tm[p * 9 + r] * ip;
          else v = Tm[c * 9 + r] - Tm[p * 9 + r] * Tm[c * 9 + p] * ip;
          nv[t] = v;
        }
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int t = 0; t < 2; t++)
        if (lane + 64 * t < 81) Tm[lane + 64 * t] = nv[t];
      __builtin_amdgcn_wave_barrier();
    }
    for (int e = lane; e < 81; e += 64) Ci[j][e] = Tm[e];
    __builtin_amdgcn_wave_barrier();
  }
  LI_STAMP(2);
  // 3. backward: Z_j = T_j^-1 (R_j - C_{j,j+1} Z_{j+1})
  for (int j = W - 1; j >= 1; j--) {
    if (j < W - 1) {
      const double* Cn = Cos + 81 * j;
#pragma unroll
      for (int t = 0; t < NT; t++) {
        const int e = lane + 64 * t;
        if (e < 9 * NC) {
          const int a = e / NC, c = e % NC;
          double s = Rr[j - 1][a][c];
#pragma unroll
          for (int k = 0; k < 9; k++) s -= Cn[k * 9 + a] * Rr[j][k][c];
          Rr[j - 1][a][c] = s;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    double nv[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int e = lane + 64 * t;
      nv[t] = 0.0;
      if (e < 9 * NC) {
        const int a = e / NC, c = e % NC;
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 9; k++) s += Ci[j][k * 9 + a] * Rr[j - 1][k][c];
        nv[t] = s;
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < NT; t++) {
      const int e = lane + 64 * t;
      if (e < 9 * NC) Rr[j - 1][e / NC][e % NC] = nv[t];
    }
    __builtin_amdgcn_wave_barrier();
  }
  LI_STAMP(3);
  // 4. Schur complement onto the poses: row i of S = H_xx (1 + u on the diagonal) - H_xy Z_x in lane i, right-hand side -g_x + H_xy z_g
  const bool row_ok = lane < n6;
  const int i = row_ok ? lane : 0, fi = i / 6, ii = i % 6;
  double A[n6 > 6 ? n6 : 7];
#pragma unroll
  for (int c = 0; c < n6; c++) A[c] = li->Hxx[(size_t)c * n6 + i];
  double hii = 0.0;
#pragma unroll
  for (int c = 0; c < n6; c++) hii = (c == i) ? A[c] : hii;
  if (fi == 0) hii = 1.0;
#pragma unroll
  for (int c = 0; c < n6; c++) A[c] = (c == i) ? A[c] + u * A[c] : A[c];
  double b = -li->g[DIM * fi + ii];
  if (fi >= 1) {
    for (int d = -1; d <= 1; d++) {
      const int j = fi + d;
      if (j < 1 || j >= W) continue;
      const double* Bb = li->B + ((size_t)fi * 3 + d + 1) * 54;   // B_{fi,j}: 6 x 9
      double bq[9];
#pragma unroll
      for (int qq = 0; qq < 9; qq++) bq[qq] = Bb[qq * 6 + ii];
#pragma unroll
      for (int c = 0; c < n6; c++) {
        double s = 0.0;
#pragma unroll
        for (int qq = 0; qq < 9; qq++) s += bq[qq] * Rr[j - 1][qq][c];
        A[c] -= s;
      }
      double s = 0.0;
#pragma unroll
      for (int qq = 0; qq < 9; qq++) s += bq[qq] * Rr[j - 1][qq][n6];
      b += s;
    }
  }
  const double gi = (fi == 0) ? 0.0 : li->g[DIM * fi + ii];
  LI_STAMP(4);
  // 5. dense solve of the 6(W-1) pose unknowns
  const double x = vxk::dense_solve_rows<n6>(A, b, colbuf, lane);
  xs[lane] = row_ok ? x : 0.0;
  __builtin_amdgcn_wave_barrier();
  LI_STAMP(5);
  // 6. dy = -z_g - Z_x dx
  for (int e = lane; e < 9 * W; e += 64) {
    const int j = e / 9, qq = e % 9;
    double s = 0.0;
    if (j >= 1) {
      s = -Rr[j - 1][qq][n6];
      for (int c = 6; c < n6; c++) s -= Rr[j - 1][qq][c] * xs[c];
    }
    ys[e] = s;
  }
  __builtin_amdgcn_wave_barrier();
  // 7. dxi, q1 = 0.5 dxi . (u D dxi - g)   (voxel_map.hpp:610-611; gauge rows contribute nothing)
  double part = row_ok ? x * (u * hii * x - gi) : 0.0;
  for (int e = lane; e < 9 * W; e += 64) {
    const int j = e / 9, qq = e % 9;
    if (j >= 1) {
      const double d = Cds[81 * j + qq * 9 + qq], y = ys[e];
      part += y * (u * d * y - gy[e]);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off);
  if (lane == 0) li->q1 = 0.5 * part;
  for (int e = lane; e < DIM * W; e += 64) {
    const int j = e / DIM, k = e % DIM;
    li->dxi[e] = k < 6 ? xs[6 * j + k] : ys[9 * j + k - 6];
  }
  LI_STAMP(6);
  // 8. trial state (voxel_map.hpp:599-606) and the factors' bias deltas (:608-609)
  if (lane < W) {
    const double* s = li->states + SL * lane;
    double* t = li->trial + SL * lane;
    double dl[DIM];
#pragma unroll
    for (int k = 0; k < 6; k++) dl[k] = xs[6 * lane + k];
#pragma unroll
    for (int k = 0; k < 9; k++) dl[6 + k] = ys[9 * lane + k];
    double Rn[9];
    vxk::lm_right_multiply_exp(s, dl, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) { t[k] = Rn[k]; lm->ctl[0].xt[12 * lane + k] = Rn[k]; }
#pragma unroll
    for (int k = 0; k < 12; k++) t[9 + k] = s[9 + k] + dl[3 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) { t[21 + k] = s[21 + k]; lm->ctl[0].xt[12 * lane + 9 + k] = s[9 + k] + dl[3 + k]; }
    if (lane < W - 1) vxi::imu_update_state(li->imus + (size_t)IL * lane, dl);
  }
  LI_STAMP(7);
#undef LI_STAMP
}

// ---- accept / reject --------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void li_decide_kernel(LIState* __restrict__ li, LMState* __restrict__ lm, const double* __restrict__ k2_partial, int nparts, int W) {
  if (li->done) return;
  const int lane = threadIdx.x;
  double s = 0.0;
  for (int k = lane; k < nparts; k += 64) s += k2_partial[k];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
  double im = 0.0;
  for (int f = 0; f < W - 1; f++) im += li->imu_res_trial[f];
  const double residual1 = li->residual1, residual2 = s + im * (li->imu_coef * 0.5);
  const double q = residual1 - residual2;
  const bool accept = q > 0;
  __builtin_amdgcn_wave_barrier();
  if (accept) {
    for (int e = lane; e < SL * W; e += 64) li->states[e] = li->trial[e];
    for (int e = lane; e < 12 * W; e += 64) lm->ctl[0].x[e] = lm->ctl[0].xt[e];
  } else if (lane < W - 1) {
    vxi::imu_rollback(li->imus + (size_t)IL * lane);
  }
  if (lane == 0) {
    double u = li->u, v = li->v;
    const double u_used = u, v_used = v, q1 = li->q1;
    if (accept) {   // voxel_map.hpp:622-636
      double rho = q / q1;
      v = 2;
      rho = 1 - (2 * rho - 1) * (2 * rho - 1) * (2 * rho - 1);
      u *= (rho < 1.0 / 3 ? 1.0 / 3 : rho);
    } else {
      u = u * v;
      v = 2 * v;
    }
    double* tr = li->trace + 8 * li->iter;
    tr[0] = residual1; tr[1] = residual2; tr[2] = u_used; tr[3] = v_used; tr[4] = q; tr[5] = q1; tr[6] = accept ? 1.0 : 0.0; tr[7] = li->calc_hess;
    li->u = u; li->v = v; li->residual2 = residual2;
    li->calc_hess = accept ? 1 : 0;
    lm->ctl[0].calc_hess = accept ? 1 : 0;
    li->iter += 1;
    if (fabs((residual1 - residual2) / residual1) < 1e-6) { li->done = 1; lm->ctl[0].done = 1; }
  }
}

__global__ __launch_bounds__(256) void li_init_kernel(LIState* __restrict__ li, LMState* __restrict__ lm, int W, double imu_coef) {
  const int t = threadIdx.x;
  if (t < 12 * W) {
    const int j = t / 12, k = t % 12;
    const double v = li->states[SL * j + k];
    lm->ctl[0].x[t] = v; lm->ctl[0].xt[t] = v;
  }
  for (int e = t; e < SL * W; e += 256) li->trial[e] = li->states[e];
  if (t == 0) {
    li->u = 0.01; li->v = 2.0; li->residual1 = 0; li->residual2 = 0; li->q1 = 0; li->imu_coef = imu_coef;
    li->calc_hess = 1; li->done = 0; li->iter = 0;
    vxk::LMCtl& c = lm->ctl[0];
    c.u = 0.01; c.v = 2.0; c.residual1 = 0; c.residual2 = 0; c.q1 = 0; c.resis[0] = 0; c.resis[1] = 0;
    c.calc_hess = 1; c.done = 0; c.iter = 0; c.converge = 1; c.rejected = 0; c.bench_mode = 0; c.n_accept = 0; c.n_reject = 0;
  }
}

// ---- launchers ----------------------------------------------------------------------------------------------------------------------
void launch_li_init(LIState* li, LMState* lm, int W, double imu_coef, hipStream_t s) { li_init_kernel<<<1, 256, 0, s>>>(li, lm, W, imu_coef); }
void launch_li_imu(LIState* li, int W, int trial, hipStream_t s) {
  if (W > 1) li_imu_kernel<<<W - 1, 64, 0, s>>>(li, W, trial);
}
void launch_li_assemble(LIState* li, const double* d_packed, int W, double* d_hess_out, hipStream_t s) {
  const int n = DIM * W, total = n * n + n + 1;
  li_assemble_kernel<<<(total + 255) / 256, 256, 0, s>>>(li, d_packed, W, d_hess_out);
}
void launch_li_solve(LIState* li, LMState* lm, int W, hipStream_t s) {
  switch (W) {
    case 1: li_solve_kernel<1><<<1, 64, 0, s>>>(li, lm); break;
    case 2: li_solve_kernel<2><<<1, 64, 0, s>>>(li, lm); break;
    case 3: li_solve_kernel<3><<<1, 64, 0, s>>>(li, lm); break;
    case 4: li_solve_kernel<4><<<1, 64, 0, s>>>(li, lm); break;
    case 5: li_solve_kernel<5><<<1, 64, 0, s>>>(li, lm); break;
    case 6: li_solve_kernel<6><<<1, 64, 0, s>>>(li, lm); break;
    case 7: li_solve_kernel<7><<<1, 64, 0, s>>>(li, lm); break;
    case 8: li_solve_kernel<8><<<1, 64, 0, s>>>(li, lm); break;
    case 9: li_solve_kernel<9><<<1, 64, 0, s>>>(li, lm); break;
    case 10: li_solve_kernel<10><<<1, 64, 0, s>>>(li, lm); break;
    default: break;
  }
}
void launch_li_decide(LIState* li, LMState* lm, const double* d_k2_partial, int nparts, int W, hipStream_t s) {
  li_decide_kernel<<<1, 64, 0, s>>>(li, lm, d_k2_partial, nparts, W);
}

}  // namespace vxli
